/*
 * refhost_wrap.cpp -- C entry points around the REFERENCE's own `Havac` class.
 *
 * TEST INFRASTRUCTURE ONLY (like oracle/_ref): nothing under havac_amd/ and nothing in bench.py's timed region loads
 * the library this file is linked into.  This file is ours; the class it drives is the reference's host/Havac.cpp,
 * compiled -- with host/phmm/PhmmPreprocessor.cpp, host/sequence/SequencePreprocessor.cpp and
 * PhmmReprojection/PhmmReprojection.cpp -- from where those files lie under /root/reference by tests/refhost/Makefile
 * into tests/_refhost/libhavac_refhost.so.  No reference source is copied into this repository.
 *
 * What the build substitutes, and nothing else:
 *   host/HavacHwClient.hpp   -> integration/HavacHwClient.hpp (the binding over include/havac_dev.h; the reference's
 *                               header needs XRT and boost).  The reference's include guard HAVAC_HW_CLIENT_HPP is
 *                               pre-defined so its body is skipped, ours is force-included.
 *   device/PublicDefines.h   -> guard HAVAC_HLS_PUBLIC_DEFINES_H pre-defined (the header includes Vitis' <ap_int.h>),
 *                               and -DNUM_CELL_PROCESSORS=12288 = NUM_CELL_GROUPS * CELLS_PER_GROUP = 16 * 768 of
 *                               device/PublicDefines.h:18-22, the one macro SequencePreprocessor.cpp takes from it.
 *   <FastaVector.h>, <p7HmmReader.h> -> the product's readers (havac_amd/csrc/host): the reference's are un-vendored
 *                               submodules, empty in the tree.  So this is a DIFFERENTIAL run: the reference's control
 *                               flow, projection arithmetic, packer (rand() order and the 'Y' quirk included) and hit
 *                               resolver over OUR readers and OUR device layer.
 */
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "Havac.hpp" /* the reference's, -I$(REFERENCE)/host */

namespace {
struct RefHost {
    Havac *havac = nullptr;
    std::vector<HavacHit> hits;
    std::string error;
};

/* exception type -> the C ABI's code for it (include/havac_dev.h), so tests compare kinds across both wrappers */
template <class F> int guarded(RefHost *h, F &&f) {
    try {
        f();
        return 0;
    } catch (const std::length_error &e) { h->error = e.what(); return -1;
    } catch (const std::logic_error &e) { h->error = e.what(); return -2;
    } catch (const std::bad_alloc &e) { h->error = e.what(); return -4;
    } catch (const std::runtime_error &e) { h->error = e.what(); return -3;
    } catch (const std::exception &e) { h->error = e.what(); return -3;
    }
}
}  // namespace

extern "C" {

int refhost_create(uint32_t device_index, float p_value, void **out) {
    RefHost *h = new RefHost();
    const int rc = guarded(h, [&] { h->havac = new Havac(device_index, p_value, "unused.xclbin"); });
    if (rc != 0) {
        delete h;
        *out = nullptr;
        return rc;
    }
    *out = h;
    return 0;
}
/* Havac::~Havac calls p7HmmListDealloc on a struct it malloc'ed and never cleared
 * (host/Havac.cpp:26,35: the FASTA struct is initialised in the ctor, the model list is not): only safe once loadPhmm has
 * run.  Tests that stop earlier leak the object. */
void refhost_destroy(void *p, int phmm_loaded) {
    RefHost *h = (RefHost *)p;
    if (!h) return;
    if (phmm_loaded) delete h->havac;
    delete h;
}
int refhost_load_phmm(void *p, const char *path) { RefHost *h = (RefHost *)p; return guarded(h, [&] { h->havac->loadPhmm(path); }); }
int refhost_load_sequence(void *p, const char *path) { RefHost *h = (RefHost *)p; return guarded(h, [&] { h->havac->loadSequence(path); }); }
int refhost_run(void *p) { RefHost *h = (RefHost *)p; return guarded(h, [&] { h->havac->runHardwareClient(); }); }
int refhost_run_async(void *p) { RefHost *h = (RefHost *)p; return guarded(h, [&] { h->havac->runHardwareClientAsync(); }); }
int refhost_wait(void *p) { RefHost *h = (RefHost *)p; return guarded(h, [&] { h->havac->waitHardwareClientAsync(); }); }
int refhost_abort(void *p) { RefHost *h = (RefHost *)p; return guarded(h, [&] { h->havac->abortHardwareClient(); }); }
int refhost_state(void *p) {
    RefHost *h = (RefHost *)p;
    int state = 0;
    const int rc = guarded(h, [&] { state = (int)h->havac->currentHardwareState(); });
    return rc ? rc : state;
}
/* getHitsFromFinishedRun (host/Havac.cpp:145-187); call with cap = 0 first to learn the count */
int refhost_get_hits(void *p, uint64_t *sequence_position, uint32_t *sequence_index, uint32_t *phmm_position,
                     uint32_t *phmm_index, uint32_t cap, uint32_t *count) {
    RefHost *h = (RefHost *)p;
    if (cap == 0) {
        const int rc = guarded(h, [&] { h->hits = h->havac->getHitsFromFinishedRun(); });
        if (rc) return rc;
    }
    *count = (uint32_t)h->hits.size();
    for (uint32_t i = 0; i < cap && i < h->hits.size(); i++) {
        sequence_position[i] = h->hits[i].sequencePosition;
        sequence_index[i] = h->hits[i].sequenceIndex;
        phmm_position[i] = h->hits[i].phmmPosition;
        phmm_index[i] = h->hits[i].phmmIndex;
    }
    return 0;
}
/* HavacHit::toString of hit i of the last fetch (host/Havac.cpp:202-207) */
int refhost_hit_to_string(void *p, uint32_t i, char *out, uint32_t cap) {
    RefHost *h = (RefHost *)p;
    if (i >= h->hits.size() || cap == 0) return -7;
    const std::string s = h->hits[i].toString();
    std::strncpy(out, s.c_str(), cap - 1);
    out[cap - 1] = 0;
    return 0;
}
const char *refhost_last_error(void *p) { return ((RefHost *)p)->error.c_str(); }
void refhost_srand(unsigned seed) { srand(seed); }

}
