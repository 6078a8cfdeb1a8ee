// p7HmmReader.cpp -- see p7HmmReader.h.  Own code; the reference's dependency is not vendored.
//
// HMMER3/[a-f] ASCII layout read here: a record starts with a
// line beginning "HMMER3/", then "TAG value" header lines, then a line
// beginning "HMM " (column captions) and the transition captions line, an
// optional "COMPO" line, the two node-0 lines (insert emissions, transitions),
// then per node three lines -- match emissions "<k> e1..eK MAP CONS RF MM CS",
// insert emissions, transitions -- and "//" closes the record.  Files hold any
// number of records back to back (benchmark/hmmDbByLength.py:11-21).
#include "p7HmmReader.h"

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

char *dupString(const std::string &s) {
    char *p = static_cast<char *>(std::malloc(s.size() + 1));
    if (p) std::memcpy(p, s.c_str(), s.size() + 1);
    return p;
}

std::vector<std::string> splitWs(const std::string &line) {
    std::vector<std::string> out;
    size_t i = 0;
    while (i < line.size()) {
        while (i < line.size() && std::isspace((unsigned char)line[i])) i++;
        size_t j = i;
        while (j < line.size() && !std::isspace((unsigned char)line[j])) j++;
        if (j > i) out.push_back(line.substr(i, j - i));
        i = j;
    }
    return out;
}

bool readLine(std::FILE *f, std::string &line) {
    line.clear();
    int c;
    bool any = false;
    while ((c = std::fgetc(f)) != EOF) {
        any = true;
        if (c == '\n') break;
        if (c != '\r') line.push_back((char)c);
    }
    return any;
}

bool parseScore(const std::string &tok, float *out) {
    if (tok == "*") { *out = INFINITY; return true; }
    char *end = nullptr;
    float v = std::strtof(tok.c_str(), &end);
    if (end == tok.c_str() || *end != '\0') return false;
    *out = v;
    return true;
}

void freeHmm(P7Hmm &h) {
    std::free(h.header.name);
    std::free(h.header.accessionNumber);
    std::free(h.model.matchEmissionScores);
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
    std::vector<P7Hmm> models;
    P7HmmReturnCode rc = p7HmmSuccess;
    std::string line;
    auto fail = [&](P7HmmReturnCode code) { rc = code; };

    while (rc == p7HmmSuccess && readLine(f, line)) {
        if (line.empty() || splitWs(line).empty()) continue;
        if (line.compare(0, 7, "HMMER3/") != 0) { fail(p7HmmFormatError); break; }
        P7Hmm h;
        std::memset(&h, 0, sizeof h);
        h.header.alphabet = P7HmmReaderAlphabetDna;
        bool haveLeng = false, haveMaxl = false, inBody = false;
        // ---- header tags ----
        while (readLine(f, line)) {
            std::vector<std::string> t = splitWs(line);
            if (t.empty()) continue;
            if (t[0] == "HMM") { inBody = true; break; }
            if (t[0] == "NAME" && t.size() >= 2) { std::free(h.header.name); h.header.name = dupString(t[1]); }
            else if (t[0] == "ACC" && t.size() >= 2) { std::free(h.header.accessionNumber); h.header.accessionNumber = dupString(t[1]); }
            else if (t[0] == "LENG" && t.size() >= 2) { h.header.modelLength = (uint32_t)std::strtoul(t[1].c_str(), nullptr, 10); haveLeng = true; }
            else if (t[0] == "MAXL" && t.size() >= 2) { h.header.maxLength = (uint32_t)std::strtoul(t[1].c_str(), nullptr, 10); haveMaxl = true; }
            else if (t[0] == "ALPH" && t.size() >= 2) {
                std::string a = t[1];
                for (auto &c : a) c = (char)std::tolower((unsigned char)c);
                h.header.alphabet = a == "amino" ? P7HmmReaderAlphabetAmino : a == "rna" ? P7HmmReaderAlphabetRna : P7HmmReaderAlphabetDna;
            } else if (t[0] == "STATS" && t.size() >= 5 && t[1] == "LOCAL") {
                float a = std::strtof(t[3].c_str(), nullptr), b = std::strtof(t[4].c_str(), nullptr);
                if (t[2] == "MSV") { h.stats.msvGumbelMu = a; h.stats.msvGumbelLambda = b; h.stats.hasMsv = 1; }
                else if (t[2] == "VITERBI") { h.stats.viterbiGumbelMu = a; h.stats.viterbiGumbelLambda = b; }
                else if (t[2] == "FORWARD") { h.stats.forwardTau = a; h.stats.forwardLambda = b; }
            }
        }
        if (!inBody || !haveLeng || h.header.modelLength == 0) { freeHmm(h); fail(p7HmmFormatError); break; }
        if (!haveMaxl) h.header.maxLength = 0;
        const uint32_t K = p7HmmGetAlphabetCardinality(&h);
        const uint32_t L = h.header.modelLength;
        h.model.matchEmissionScores = static_cast<float *>(std::malloc(sizeof(float) * (size_t)L * K));
        if (!h.model.matchEmissionScores) { freeHmm(h); fail(p7HmmAllocationFailure); break; }
        // ---- body ----
        bool ok = readLine(f, line);                 // transition captions
        ok = ok && readLine(f, line);                // COMPO or node-0 insert emissions
        if (ok) {
            std::vector<std::string> t = splitWs(line);
            if (!t.empty() && t[0] == "COMPO") ok = readLine(f, line);   // -> node-0 insert emissions
        }
        ok = ok && readLine(f, line);                // node-0 transitions
        for (uint32_t k = 1; ok && k <= L; k++) {
            ok = readLine(f, line);
            std::vector<std::string> t = ok ? splitWs(line) : std::vector<std::string>();
            if (!ok || t.size() < 1 + K || std::strtoul(t[0].c_str(), nullptr, 10) != k) { ok = false; break; }
            for (uint32_t a = 0; a < K; a++)
                if (!parseScore(t[1 + a], &h.model.matchEmissionScores[(size_t)(k - 1) * K + a])) ok = false;
            ok = ok && readLine(f, line);            // insert emissions
            ok = ok && readLine(f, line);            // transitions
        }
        bool closed = false;
        while (ok && readLine(f, line)) {
            std::vector<std::string> t = splitWs(line);
            if (t.empty()) continue;
            closed = (t[0] == "//");
            break;
        }
        if (!ok || !closed) { freeHmm(h); fail(p7HmmFormatError); break; }
        models.push_back(h);
    }
    std::fclose(f);
    if (rc == p7HmmSuccess && models.empty()) rc = p7HmmFormatError;
    if (rc != p7HmmSuccess) {
        for (auto &m : models) freeHmm(m);
        return rc;
    }
    list->phmms = static_cast<P7Hmm *>(std::malloc(sizeof(P7Hmm) * models.size()));
    if (!list->phmms) {
        for (auto &m : models) freeHmm(m);
        return p7HmmAllocationFailure;
    }
    std::memcpy(list->phmms, models.data(), sizeof(P7Hmm) * models.size());
    list->count = (uint32_t)models.size();
    return p7HmmSuccess;
}

}  // extern "C"
