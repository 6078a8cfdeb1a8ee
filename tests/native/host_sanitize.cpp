// Host stages under AddressSanitizer / UBSan (CPU build only; tests/test_sanitizers.py compiles and runs this).
// Reads every file named on the command line with the readers of havac_amd/csrc/host and pushes the result through the
// packers, the reverse-strand builder, the patch collector, the position lookup and both model preprocessors.
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "FastaVector.h"
#include "PhmmPreprocessor.hpp"
#include "SequencePreprocessor.hpp"
#include "p7HmmReader.h"

int main(int argc, char **argv) {
    for (int i = 1; i < argc; i++) {
        const std::string path = argv[i];
        if (path.size() > 3 && path.substr(path.size() - 3) == ".fa") {
            FastaVector fv;
            fastaVectorInit(&fv);
            const int rc = fastaVectorReadFasta(path.c_str(), &fv);
            std::printf("%s: rc %d, %zu chars, %zu records\n", path.c_str(), rc, fv.sequence.count, fv.metadata.count);
            if (rc == FASTA_VECTOR_OK) {
                std::srand(1);
                SequencePreprocessor plain(&fv);
                SequencePreprocessor separated(&fv, true);
                std::vector<uint64_t> starts, residues;
                for (size_t j = 0; j < fv.metadata.count; j++) {
                    const size_t begin = j ? fv.metadata.data[j - 1].sequenceEndPosition : 0;
                    starts.push_back(begin);
                    residues.push_back(fv.metadata.data[j].sequenceEndPosition - begin - 1);
                }
                plain.appendReverseStrand(starts, residues);
                std::vector<uint64_t> columns;
                std::vector<uint8_t> symbols;
                std::srand(1);
                SequencePreprocessor::collectPatches(&fv, columns, symbols);
                for (size_t g = 0; g <= fv.sequence.count + 3; g += 7) {
                    FastaVectorLocalPosition where;
                    fastaVectorGetLocalSequencePositionFromGlobal(&fv, g, &where);
                }
                std::printf("  packed %zu bytes, separated %zu bytes, %zu patches\n", plain.getCompressedSequenceBuffer().size(),
                            separated.getCompressedSequenceBuffer().size(), columns.size());
            }
            fastaVectorDealloc(&fv);
        } else {
            P7HmmList list;
            const int rc = readP7Hmm(path.c_str(), &list);
            std::printf("%s: rc %d\n", path.c_str(), rc);
            if (rc == p7HmmSuccess) {
                try {
                    PhmmPreprocessor probe(&list, 0.02f);
                } catch (const std::exception &e) {
                    std::printf("  rejected: %s\n", e.what());
                    p7HmmListDealloc(&list);
                    continue;
                }
                PhmmPreprocessor plain(&list, 0.02f);
                PhmmPreprocessor separated(&list, 0.02f, true);
                std::printf("  %u models, %zu bytes, separated %zu bytes\n", list.count, plain.getProcessedPhmmData()->size(),
                            separated.getProcessedPhmmData()->size());
                p7HmmListDealloc(&list);
            }
        }
    }
    return 0;
}
