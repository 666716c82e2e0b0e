/*
 * bsq_diag.h -- measurement and diagnostic exports of libbsq_hip.so.  NOT part of the drop-in surface (bsq.h):
 * nothing here is needed to use the tokenizer; bench.py, scripts/ and the tests use it to select kernel variants
 * for A/B runs, to measure the write-bandwidth yardsticks the roofline numbers are quoted against, and to run the
 * in-library self-checks.  Results of the product entry points never depend on any of it.
 */
#ifndef BSQ_DIAG_H
#define BSQ_DIAG_H

#include "bsq.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Build id of this library: the first 16 hex digits of sha256 over every source and header of csrc/ + include/ and the compiler
 * flags, computed by bioseq_amd/build.py and compiled in.  profiles/traffic.json records the id its counters were measured on;
 * bench.py prints `traffic_stale: true` when the loaded library's differs (the traffic figure in the line is then an old build's). */
const char *bsq_build_id(void);

/* Tuning knobs (kernel variants for A/B measurements; results never change).  Names and meaning:
 * bioseq_amd/csrc/bsq_internal.h.  Environment variables BSQ_<NAME> give the initial values. */
bsq_status bsq_tuning_set(const char *name, int32_t value);
int32_t bsq_tuning_get(const char *name);

/* Bytes the *_host entry points have copied host -> device since the library was loaded (the device entry points copy
 * nothing).  The loader tests assert that a shuffled epoch over a resident FlatFile leaves it unchanged. */
uint64_t bsq_host_upload_bytes(void);
/* the augmentation kernel's integer acceptance thresholds (row r of normrows: a position is accepted iff lo32 <= out21[r]);
 * tests check them against the floating-point test of the numpy twin at every boundary */
bsq_status bsq_blosum62_accept_thresholds(uint32_t *out21);

/* Streaming fill of nbytes (multiple of 16, 16-byte aligned) with a 32-bit pattern: the
 * write-bandwidth yardstick bench.py reports next to the encode kernels. */
bsq_status bsq_fill_device(void *dst, size_t nbytes, uint32_t pattern, void *hip_stream);
/* Writes a (rows x pitch bytes) matrix with the tiled one-hot kernel's store pattern and none
 * of its work -- block (cb, rb) owns `seg` contiguous bytes of 4*rows_per_wave rows, one wave per
 * rows_per_wave rows.  Used by scripts/sweep_pattern.py to separate pattern cost from kernel cost. */
bsq_status bsq_fill_pattern_device(void *dst, int64_t rows, int64_t pitch, int32_t seg, int32_t rows_per_wave,
                                   int32_t order, int32_t interleave, int32_t nt, void *hip_stream);

/* Read + write stream of the (B,P) token kernels' shape with none of their work: wave k writes the aligned 4-KiB chunk k
 * of dst (dst_bytes a multiple of 4096) and reads its share of src (src_bytes <= dst_bytes), coalesced 16-byte pieces.
 * mode 0: loads then stores of the loaded data; 1: an extra dependent load first (offsets -> characters); 2: stores
 * that do not wait for the loads; 3: the loads only (no store traffic).  The yardstick bench.py quotes cfg2 / cfg5 against.
 * Round 5, persistent workgroups (knob fill_mode = workgroups per CU, default 4): mode 4: every wave walks its XCD class's chunks with the
 * loads of its next chunk issued before the stores of the current one; modes 5 / 6 / 7: one LDS-DMA loader wave (global_load_lds_dwordx4)
 * stages the pieces of three consumer waves 1 / 2 / 3 steps ahead through an LDS ring.  (profiles/r05/persistent_stream_lab.txt) */
bsq_status bsq_copy_mix_device(void *dst, size_t dst_bytes, const void *src, size_t src_bytes, int32_t mode, int32_t nt,
                               void *hip_stream);

/* Host-only self-test of the kernels' division-free index arithmetic (reciprocal multiplies instead of integer
 * divisions; the same inline functions run on the device): 0 = every case exact. */
int64_t bsq_selftest_index_math(void);

/* xcd_dev[b] = id (0..7, HW_REG_XCC_ID) of the XCD block b of an nblocks-block 1-D launch ran on.  The chunk
 * kernels assume -- for speed only, never for results -- that blocks b and b + 8 share an XCD. */
bsq_status bsq_xcd_of_blocks_device(int32_t *xcd_dev, int32_t nblocks, void *hip_stream);
/* 1 if a probe launch on the current device found the round-robin placement the chunk kernels are tuned for
 * (blocks b and b + 8 on one XCD, 8 consecutive blocks on 8 different XCDs), 0 if not, -1 on error.  Probed once
 * per device and cached. */
int32_t bsq_xcd_round_robin(void);

#ifdef __cplusplus
}
#endif
#endif /* BSQ_DIAG_H */
