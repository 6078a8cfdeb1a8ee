// p7HmmReader.cpp -- see p7HmmReader.h.  Own code; the reference's dependency is not vendored.
//
// HMMER3/[a-f] ASCII layout read here: a record starts with a
// line beginning "HMMER3/", then "TAG value" header lines, then a line
// beginning "HMM " (column captions) and the transition captions line, an
// optional "COMPO" line, the two node-0 lines (insert emissions, transitions),
// then per node three lines -- match emissions "<k> e1..eK MAP CONS RF MM CS",
// insert emissions, transitions -- and "//" closes the record.  Files hold any
// number of records back to back (benchmark/hmmDbByLength.py:11-21).
//
// The file is read into memory once, cut into records at the lines that begin "HMMER3/", and the records are parsed
// side by side on the host's cores (a Dfam-sized file is 30+ MB of text: round 2's line-by-line reader with strtof
// took 0.3 s for 150,000 model positions, most of the end-to-end time of a run).  Of a node's three lines only the
// match emissions are tokenised; the other two are skipped to their newline.  Scores written as plain decimals of up
// to seven significant digits -- what hmmbuild writes -- are converted as (float)digits / 10^k, one IEEE division of two
// exactly representable floats, hence the correctly rounded value strtof would return; anything else goes to strtof.
#include "p7HmmReader.h"

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "HostThreads.hpp"

namespace {

struct Span { const char *b, *e; };     // [b, e)

inline bool isBlank(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

// the next line of [p, end): *line = its text without the newline, returns false at the end of the text
inline bool nextLine(const char *&p, const char *end, Span *line) {
    if (p >= end) return false;
    const char *nl = static_cast<const char *>(std::memchr(p, '\n', (size_t)(end - p)));
    line->b = p;
    line->e = nl ? nl : end;
    p = nl ? nl + 1 : end;
    return true;
}

// the next whitespace-separated token of a line; false when there is none
inline bool nextToken(Span &line, Span *tok) {
    const char *p = line.b;
    while (p < line.e && isBlank(*p)) p++;
    if (p >= line.e) { line.b = p; return false; }
    const char *q = p;
    while (q < line.e && !isBlank(*q)) q++;
    tok->b = p; tok->e = q;
    line.b = q;
    return true;
}

inline bool tokenIs(const Span &t, const char *s) {
    const size_t n = std::strlen(s);
    return (size_t)(t.e - t.b) == n && std::memcmp(t.b, s, n) == 0;
}

inline std::string tokenString(const Span &t) { return std::string(t.b, t.e); }

char *dupToken(const Span &t) {
    const size_t n = (size_t)(t.e - t.b);
    char *p = static_cast<char *>(std::malloc(n + 1));
    if (p) { std::memcpy(p, t.b, n); p[n] = '\0'; }
    return p;
}

// A whole token as a float, as strtof reads it (false: not a number, or trailing characters).
bool tokenToFloat(const Span &t, float *out) {
    static const float kPow10[11] = {1e0f, 1e1f, 1e2f, 1e3f, 1e4f, 1e5f, 1e6f, 1e7f, 1e8f, 1e9f, 1e10f};   // all exact in binary32
    const char *p = t.b;
    bool negative = false;
    if (p < t.e && (*p == '-' || *p == '+')) { negative = *p == '-'; p++; }
    // the plain form: digits [. digits], at most seven significant digits (so the digits as an integer are below 2^24)
    uint32_t mantissa = 0;
    int significant = 0, fraction = 0;
    bool plain = true, point = false, anyDigit = false;
    for (; p < t.e; p++) {
        const char c = *p;
        if (c >= '0' && c <= '9') {
            anyDigit = true;
            if (mantissa != 0 || c != '0') significant++;
            if (significant > 7) { plain = false; break; }
            mantissa = mantissa * 10u + (uint32_t)(c - '0');
            if (point) fraction++;
        } else if (c == '.' && !point) {
            point = true;
        } else {
            plain = false;
            break;
        }
    }
    if (plain && anyDigit && fraction <= 10) {
        // (float)mantissa and 10^fraction are exact; IEEE division rounds their quotient once: the correctly rounded value
        const float v = (float)mantissa / kPow10[fraction];
        *out = negative ? -v : v;
        return true;
    }
    char buf[64];
    const size_t n = (size_t)(t.e - t.b);
    if (n == 0 || n >= sizeof buf) return false;
    std::memcpy(buf, t.b, n);
    buf[n] = '\0';
    char *end = nullptr;
    const float v = std::strtof(buf, &end);
    if (end == buf || *end != '\0') return false;
    *out = v;
    return true;
}

inline unsigned long tokenToUnsigned(const Span &t) {      // strtoul's reading of the token's leading digits
    unsigned long v = 0;
    for (const char *p = t.b; p < t.e && *p >= '0' && *p <= '9'; p++) v = v * 10ul + (unsigned long)(*p - '0');
    return v;
}

inline float tokenToFloatLoose(const Span &t) {            // strtof(..., nullptr): whatever prefix parses, 0 otherwise
    char buf[64];
    const size_t n = std::min((size_t)(t.e - t.b), sizeof buf - 1);
    std::memcpy(buf, t.b, n);
    buf[n] = '\0';
    return std::strtof(buf, nullptr);
}

bool parseScore(const Span &tok, float *out) {
    if (tok.e - tok.b == 1 && *tok.b == '*') { *out = INFINITY; return true; }
    return tokenToFloat(tok, out);
}

void freeHmm(P7Hmm &h) {
    std::free(h.header.name);
    std::free(h.header.accessionNumber);
    std::free(h.model.matchEmissionScores);
    std::memset(&h, 0, sizeof h);
}

// One record: text [b, e) that starts with its "HMMER3/" line.
P7HmmReturnCode parseRecord(const char *b, const char *e, P7Hmm &h) {
    std::memset(&h, 0, sizeof h);
    h.header.alphabet = P7HmmReaderAlphabetDna;
    const char *p = b;
    Span line, tok;
    nextLine(p, e, &line);                                   // the "HMMER3/..." line
    bool haveLeng = false, haveMaxl = false, inBody = false;
    // ---- header tags ----
    while (nextLine(p, e, &line)) {
        if (!nextToken(line, &tok)) continue;
        if (tokenIs(tok, "HMM")) { inBody = true; break; }
        Span value;
        const bool hasValue = nextToken(line, &value);
        if (!hasValue) continue;
        if (tokenIs(tok, "NAME")) { std::free(h.header.name); h.header.name = dupToken(value); }
        else if (tokenIs(tok, "ACC")) { std::free(h.header.accessionNumber); h.header.accessionNumber = dupToken(value); }
        else if (tokenIs(tok, "LENG")) { h.header.modelLength = (uint32_t)tokenToUnsigned(value); haveLeng = true; }
        else if (tokenIs(tok, "MAXL")) { h.header.maxLength = (uint32_t)tokenToUnsigned(value); haveMaxl = true; }
        else if (tokenIs(tok, "ALPH")) {
            std::string a = tokenString(value);
            for (auto &c : a) c = (char)std::tolower((unsigned char)c);
            h.header.alphabet = a == "amino" ? P7HmmReaderAlphabetAmino : a == "rna" ? P7HmmReaderAlphabetRna : P7HmmReaderAlphabetDna;
        } else if (tokenIs(tok, "STATS") && tokenIs(value, "LOCAL")) {
            Span kind, first, second;
            if (nextToken(line, &kind) && nextToken(line, &first) && nextToken(line, &second)) {
                const float x = tokenToFloatLoose(first), y = tokenToFloatLoose(second);
                if (tokenIs(kind, "MSV")) { h.stats.msvGumbelMu = x; h.stats.msvGumbelLambda = y; h.stats.hasMsv = 1; }
                else if (tokenIs(kind, "VITERBI")) { h.stats.viterbiGumbelMu = x; h.stats.viterbiGumbelLambda = y; }
                else if (tokenIs(kind, "FORWARD")) { h.stats.forwardTau = x; h.stats.forwardLambda = y; }
            }
        }
    }
    if (!inBody || !haveLeng || h.header.modelLength == 0) { freeHmm(h); return p7HmmFormatError; }
    if (!haveMaxl) h.header.maxLength = 0;
    const uint32_t K = p7HmmGetAlphabetCardinality(&h);
    const uint32_t L = h.header.modelLength;
    // a record cannot hold more nodes than it has lines: refuse a LENG that promises more before allocating for it
    if ((size_t)L > (size_t)(e - p)) { freeHmm(h); return p7HmmFormatError; }
    h.model.matchEmissionScores = static_cast<float *>(std::malloc(sizeof(float) * (size_t)L * K));
    if (!h.model.matchEmissionScores) { freeHmm(h); return p7HmmAllocationFailure; }
    // ---- body ----
    bool ok = nextLine(p, e, &line);                         // transition captions
    ok = ok && nextLine(p, e, &line);                        // COMPO or node-0 insert emissions
    if (ok) {
        Span first = line;
        if (nextToken(first, &tok) && tokenIs(tok, "COMPO")) ok = nextLine(p, e, &line);   // -> node-0 insert emissions
    }
    ok = ok && nextLine(p, e, &line);                        // node-0 transitions
    for (uint32_t k = 1; ok && k <= L; k++) {
        ok = nextLine(p, e, &line);
        if (!ok || !nextToken(line, &tok) || tokenToUnsigned(tok) != k) { ok = false; break; }
        float *scores = &h.model.matchEmissionScores[(size_t)(k - 1) * K];
        for (uint32_t a = 0; a < K; a++) {
            if (!nextToken(line, &tok)) { ok = false; break; }
            if (!parseScore(tok, &scores[a])) ok = false;    // (the remaining tokens are still counted, as before)
        }
        if (!ok) break;
        ok = nextLine(p, e, &line);                          // insert emissions
        ok = ok && nextLine(p, e, &line);                    // transitions
    }
    bool closed = false;
    while (ok && nextLine(p, e, &line)) {
        if (!nextToken(line, &tok)) continue;
        closed = tokenIs(tok, "//");
        break;
    }
    // nothing but blank lines may follow "//" inside this record's text (the next record starts at its "HMMER3/" line)
    while (ok && closed && nextLine(p, e, &line))
        if (nextToken(line, &tok)) { ok = false; break; }
    if (!ok || !closed) { freeHmm(h); return p7HmmFormatError; }
    return p7HmmSuccess;
}

}  // namespace

extern "C" {

uint32_t p7HmmGetAlphabetCardinality(const P7Hmm *phmm) {
    return phmm->header.alphabet == P7HmmReaderAlphabetAmino ? 20u : 4u;
}

void p7HmmListDealloc(P7HmmList *list) {
    if (!list) return;
    for (uint32_t i = 0; i < list->count; i++) freeHmm(list->phmms[i]);
    std::free(list->phmms);
    list->phmms = nullptr;
    list->count = 0;
}

P7HmmReturnCode readP7Hmm(const char *path, P7HmmList *list) {
    list->phmms = nullptr;
    list->count = 0;
    std::FILE *f = std::fopen(path, "rb");
    if (!f) return p7HmmFileNotFound;
    std::vector<char> text;
    {
        if (std::fseek(f, 0, SEEK_END) == 0) {
            const long size = std::ftell(f);
            if (size > 0) text.reserve((size_t)size);
            std::rewind(f);
        }
        char buf[1 << 16];
        size_t got;
        try {
            while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) text.insert(text.end(), buf, buf + got);
        } catch (const std::bad_alloc &) {
            std::fclose(f);
            return p7HmmAllocationFailure;
        }
    }
    std::fclose(f);
    // ---- cut into records: every line that begins "HMMER3/" starts one; before the first only blank lines may stand ----
    const char *const begin = text.data(), *const end = text.data() + text.size();
    std::vector<const char *> starts;
    {
        const char *p = begin;
        Span line, tok;
        bool leadingGarbage = false;
        while (p < end) {
            const char *at = p;
            nextLine(p, end, &line);
            if (line.e - line.b >= 7 && std::memcmp(line.b, "HMMER3/", 7) == 0) starts.push_back(at);
            else if (starts.empty() && nextToken(line, &tok)) { leadingGarbage = true; break; }
        }
        if (leadingGarbage || starts.empty()) return p7HmmFormatError;
    }
    const size_t n = starts.size();
    P7Hmm *models = static_cast<P7Hmm *>(std::calloc(n, sizeof(P7Hmm)));
    if (!models) return p7HmmAllocationFailure;
    std::vector<P7HmmReturnCode> codes(n, p7HmmSuccess);
    // small files are not worth a thread start (2 us per model position against ~50 us per thread)
    const unsigned threads = text.size() < (1u << 18) ? 1u : havacHostThreads(n);
    havacParallelFor(n, threads, [&](size_t i) { codes[i] = parseRecord(starts[i], i + 1 < n ? starts[i + 1] : end, models[i]); });
    for (size_t i = 0; i < n; i++)
        if (codes[i] != p7HmmSuccess) {                     // the first bad record decides, as a sequential reader would
            const P7HmmReturnCode rc = codes[i];
            for (size_t j = 0; j < n; j++) freeHmm(models[j]);
            std::free(models);
            return rc;
        }
    list->phmms = models;
    list->count = (uint32_t)n;
    return p7HmmSuccess;
}

}  // extern "C"
