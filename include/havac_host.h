/*
 * havac_host.h -- C entry points around the C++ `Havac` class (libhavac.so).
 *
 * The reference's public API is the C++ class in host/Havac.hpp:42-107; C++
 * callers include havac_amd/csrc/host/Havac.hpp and use it directly.  These
 * wrappers exist so that non-C++ callers (the pytest suite through ctypes, or
 * any FFI) can drive the very same object, and so that the host-only stages
 * (FASTA packing, model projection, hit resolution) can be tested without a GPU.
 * Exceptions are turned into negative HAVAC_E_* codes (include/havac_dev.h).
 */
#ifndef HAVAC_HOST_H
#define HAVAC_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct havac_host havac_host;

/* Havac::Havac(deviceIndex, requiredPValue, xclbinSrc)  host/Havac.cpp:20-31 */
int havac_host_create(uint32_t device_index, float required_p_value, havac_host **out);
/* Havac(Havac::DeferredStart, ...) (an addition): returns at once, the device layer starts on a helper thread under the
 * caller's next calls; a device that cannot be opened is reported by the first call that needs it (HAVAC_E_RUNTIME). */
int havac_host_create_deferred(uint32_t device_index, float required_p_value, havac_host **out);
/* Havac over several GPUs of one node, one column shard each (an addition; include/havac_dev.h) */
int havac_host_create_multi(const uint32_t *device_indices, uint32_t ndevices, float required_p_value, havac_host **out);
void havac_host_destroy(havac_host *h);
int havac_host_load_sequence(havac_host *h, const char *fasta_path);   /* Havac::loadSequence  :57-77 */
int havac_host_load_phmm(havac_host *h, const char *hmm_path);         /* Havac::loadPhmm      :42-55 */
int havac_host_run(havac_host *h);                                     /* runHardwareClient    :80-83 */
int havac_host_run_async(havac_host *h);                               /* runHardwareClientAsync :85-94 */
int havac_host_wait(havac_host *h);                                    /* waitHardwareClientAsync :96-98 */
int havac_host_abort(havac_host *h);                                   /* abortHardwareClient  :100-102 */
int havac_host_state(havac_host *h);                                   /* currentHardwareState :190-192 */
int havac_host_set_hit_capacity(havac_host *h, uint64_t max_hits);
/* Havac::setPipelineDepth: runs in flight (not in the reference; 1 = one run at a time, as there) */
int havac_host_set_pipeline_depth(havac_host *h, uint32_t depth);
/* with several runs open: the caller is done with the hits havac_host_get_hits served (it keeps a copy, for the count-then-arrays
 * pair of calls); the next havac_host_get_hits fetches the next run.  Nothing to do at depth 1. */
int havac_host_next_run(havac_host *h);
/* Havac::setBoundaryMode (not in the reference): score every (model, record) pair on its own; before the loads */
int havac_host_set_boundary_mode(havac_host *h, int on);
/* Havac::setBothStrands (not in the reference): also score every record's reverse complement; before loadSequence */
int havac_host_set_both_strands(havac_host *h, int on);
/* Havac::setDevicePacking (not in the reference; on by default): pack the text on the GPU; before loadSequence */
int havac_host_set_device_packing(havac_host *h, int on);
/* reverseStrand flag of the hits of the last havac_host_get_hits call, one byte each */
int havac_host_get_hit_strands(havac_host *h, uint8_t *reverse, uint32_t cap, uint32_t *count);
/* Havac::getHitsFromFinishedRun :145-187.  First call with cap = 0 to learn the count. */
int havac_host_get_hits(havac_host *h, uint64_t *sequence_position, uint32_t *sequence_index,
                        uint32_t *phmm_position, uint32_t *phmm_index, uint32_t cap, uint32_t *count);
/* the raw 64-bit records the last get_hits call fetched from the device */
int havac_host_get_raw_hits(havac_host *h, uint64_t *out, uint32_t cap, uint32_t *count);
int havac_host_last_run_ms(havac_host *h, float *ssv_kernel_ms, float *total_ms);
const char *havac_host_last_error(havac_host *h);

/* ---- host-only stages, no device needed -------------------------------- */

/* FastaVector + SequencePreprocessor: file -> 2-bit packed, segment-padded bytes.
 * `seed` >= 0 calls srand(seed) first (ambiguity codes and record terminators
 * draw from rand(), host/sequence/SequencePreprocessor.cpp:74,82). */
int havac_host_pack_fasta(const char *fasta_path, int64_t seed, uint8_t *out, uint64_t cap, uint64_t *nbytes,
                          uint64_t *nchars, uint32_t *nrecords);

/* P7HmmReader + PhmmPreprocessor: file -> concatenated int8 [row][A,C,G,T]. */
int havac_host_project_hmm(const char *hmm_path, float p_value, int8_t *out, uint64_t cap, uint64_t *nbytes,
                           uint32_t *nmodels, uint32_t *model_lengths, uint32_t lengths_cap);

/* P7HmmReader alone: the match-emission file values (-ln p; +inf for '*') of every model of the file, back to back,
 * [position][A,C,G,T].  Writes at most `cap` floats; *nvalues = the full count.  (Checking aid for the reader.) */
int havac_host_read_hmm_emissions(const char *hmm_path, float *out, uint64_t cap, uint64_t *nvalues, uint32_t *nmodels);

/* p7HmmProjectForThreshold256 on explicit parameters (PhmmReprojection.cpp:109-145): `emissions` = model_length x 4
 * match-emission file values (-ln p, +inf for '*'), `out` = model_length x 4 int8. */
int havac_host_project_model(float mu, float lambda, uint32_t max_length, uint32_t model_length, float p_value,
                             const float *emissions, int8_t *out);
/* esl_gumbel_invsurv (PhmmReprojection.cpp:15-31). */
double havac_host_gumbel_invsurv(double p, double mu, double lambda);
/* findThreshold256ScalingFactor on explicit parameters (PhmmReprojection.cpp:36-64). */
float havac_host_scaling_factor(float mu, float lambda, uint32_t max_length, uint32_t model_length, float p_value);
/* emissionScoreToProjectedScore (PhmmReprojection.cpp:90-107). */
float havac_host_project_score(float emission_score, float multiplier);

/* The resolver alone: raw records + the two input files -> HavacHit fields. */
int havac_host_resolve_hits(const char *fasta_path, const char *hmm_path, const uint64_t *raw, uint32_t nraw,
                            uint64_t *sequence_position, uint32_t *sequence_index, uint32_t *phmm_position,
                            uint32_t *phmm_index, uint32_t cap, uint32_t *count);

/* The host packer in its other layouts (checking aid for the GPU-built ones): boundary mode (SequencePreprocessor(fasta,
 * true): packed sequence + separator bitmap) and/or both strands (appendReverseStrand).  Writes at most `cap` / `mask_cap`
 * bytes; *nbytes / *mask_nbytes = the full sizes (mask 0 without boundary mode); *forward_columns = the forward half's
 * columns (0 without both strands). */
int havac_host_pack_fasta_layout(const char *fasta_path, int64_t seed, int boundary_mode, int both_strands, uint8_t *out,
                                 uint64_t cap, uint64_t *nbytes, uint8_t *mask_out, uint64_t mask_cap, uint64_t *mask_nbytes,
                                 uint64_t *forward_columns);

/* What loadSequence sends to the GPU when it packs there: the file's characters (records + terminators; nchars) and
 * the columns that are not a/c/g/t with the symbols drawn for them (SequencePreprocessor::collectPatches); seed as
 * in havac_host_pack_fasta.  Writes at most `cap` characters / patches; *nchars, *npatches = the full counts. */
int havac_host_text_and_patches(const char *fasta_path, int64_t seed, char *chars, uint64_t chars_cap, uint64_t *nchars,
                                uint64_t *patch_columns, uint8_t *patch_symbols, uint64_t patch_cap, uint64_t *npatches);

/* Hits -> merged windows (Havac.hpp: havacMergeHitsToWindows).  `reverse_strand` may
 * be NULL (all forward).  Writes at most `cap` windows into the seven output arrays, *count = windows found. */
int havac_host_merge_windows(const uint64_t *sequence_position, const uint32_t *sequence_index,
                             const uint32_t *phmm_position, const uint32_t *phmm_index, const uint8_t *reverse_strand,
                             uint32_t nhits, const uint32_t *model_lengths, uint32_t nmodels,
                             const uint64_t *record_lengths, uint32_t nrecords, uint32_t flank,
                             uint32_t *w_sequence_index, uint32_t *w_phmm_index, uint8_t *w_reverse_strand,
                             uint64_t *w_start, uint64_t *w_end, uint32_t *w_phmm_first, uint32_t *w_phmm_last,
                             uint32_t *w_hit_count, uint32_t cap, uint32_t *count);
/* The same on a handle's finished run (Havac::getWindowsFromFinishedRun). */
int havac_host_get_windows(havac_host *h, uint32_t flank, uint32_t *w_sequence_index, uint32_t *w_phmm_index,
                           uint8_t *w_reverse_strand, uint64_t *w_start, uint64_t *w_end, uint32_t *w_phmm_first,
                           uint32_t *w_phmm_last, uint32_t *w_hit_count, uint32_t cap, uint32_t *count);

#ifdef __cplusplus
}
#endif
#endif
