// Internal helpers shared by the translation units of libbsq_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime_api.h>

#include <mutex>

#include "bsq.h"

namespace bsq_internal {

// Record a thread-local error message (returned by bsq_last_error()) and pass the status through.
bsq_status set_error(bsq_status st, const char *msg);
bsq_status set_hip_error(const char *what, hipError_t e);
// Tuning / diagnostic knobs (bsq_tuning_set, or environment BSQ_<NAME> read once, at the first launch):
// (speed only -- results never depend on them; every variant is covered by the GPU parity tests.  The knobs of experiments that lost
//  -- chunks_cpw, tokenize_nch, expand_mode, xcd_claim, chunk_math, tokens8_abl, tokens8_ring, augment_mode, raw_mode 2 / 3 -- left the
//  library in round 6 together with their kernels: csrc/labs/README.md)
//   nt_stores     1: `global_store ... nt` for the output streams (default 1)
//   onehot_path   0: automatic, 1: tiled kernel, 2: two-pass (tokens + expansion), 3: chunk-owner kernel
//   expand_pad    unused dynamic LDS of k_expand_chunks = occupancy cap: 0 automatic, > 0 bytes, < 0 none
//   expand_slots  4: k_expand_chunks always issues four token loads per step (0: as many as the chunk needs)
//   chunks_pad    the same cap for k_onehot_chunks (0: 22528 bytes = 4 workgroups per CU)
//   tokenize_path 1: never use k_tokenize_chunks / the raw-token kernel for batch_tokenize
//   tokenize_pad  unused dynamic LDS of k_tokenize_chunks (experiments: no cap helps it)
//   tile_group    G > 1: XCD-aware tile order in groups of 8 x G sequence tiles (G = 16..64: +1 % on cfg4 int8; >= 128 loses
//                 the L2 reuse: cfg4 f32 0.69 -> 0.90 ms at 512; profiles/r02/tile_lab3.txt)
//   bcl_path      channels-first one-hot: 0 automatic (two-pass k_tokens_bp8<raw> + k_expand_bcl for outputs >= 256 MB),
//                 1 never two-pass, 2 two-pass whenever it applies;  bcl_pad  occupancy cap of k_expand_bcl (0: 3 workgroups per CU)
//   raw_mode      1 / 4: force the 256 x 64 / the wide 1024 x 16 tile of k_tokens_raw (0: wide for the final (P,B)
//                 int8 matrix of >= 4096 sequences, 256 x 64 for the expansion scratch); any value != 0 also keeps the raw-id pass
//                 out of k_tokens_pb8_fast
//   tokens8       1: never use k_tokens_bp8 for the (B,P) int8 token matrix (falls back to k_tokenize_chunks / _rows);
//                 2: only for padlen % 16 == 0 and 16-byte aligned outputs (no row-piece form)
//   tokens8_lookup  0 automatic, 1 LDS byte table, 2 register table (v_perm_b32)
//   tokens8_pad   unused dynamic LDS of k_tokens_bp8
//   onehot_tb     0: automatic, else force 64 / 128 / 256 sequences per tile of k_onehot_tile
//   tile_order    0: automatic (XCD-aware), 1: position-tile index fastest, 2: XCD-aware, 3: sequence-tile index fastest,
//                 4: every XCD walks its own contiguous range of sequence tiles (automatic for (P,B) tokens of 2- / 4-byte
//                 elements whose rows are not 64-byte aligned), 5: as 4 with the position tiles of a sequence tile back to
//                 back (automatic for the (P,B) int8 token matrix)
//   fill_mode, fill_pad   access pattern / occupancy of bsq_fill_device (write-bandwidth yardsticks)
//   tokenize_tb           sequences per tile of k_tokenize_tile ((P,B) tokens of 2- / 4- / 8-byte elements): 64 / 128 / 256
//   wide_index            1: the (B,P) chunk kernels take their 64-bit index arithmetic whatever the size (tests: the path
//                         otherwise needs more than 2^31 16-byte pieces of output)
//   pattern_wait          bsq_fill_pattern_device: n > 0 = s_waitcnt vmcnt(n - 1) after every row of a wave
//   host_copy_threads     worker threads of the pipelined device -> host result copy (0: 8)
//   tokens_pb8            1: never use k_tokens_pb8_fast for the (P,B) token matrices (k_tokens_raw / k_tokenize_tile instead), 2: also
//                         for 4-byte elements (automatic: 1-, 2- and 8-byte elements), 3: only when rows and output are 16-byte aligned;
//                         its lookup follows tokens8_lookup
//   augment_fused         bsq_augment_tokenize_device: 0 automatic -- up to 16 384 chunks the no-wait form (augmentation + tokens in one launch, nobody
//                         waits, a patch launch behind it: round 5; 4: at every size), beyond that the flag form --, 1 never fused (the two launches), 2 the flag form of rounds 3-4 (token waves wait for their rows' augmentation;
//                         written-through stores, any XCD), 3 the flag form with the hand-off inside one XCD where it applies
//   expand_gate           k_expand_chunks: one pacing load in front of every wave (0 automatic: rows of 24 ... 63 bytes; 1 never; 2 always)
//   raw_nibbles           the id scratch of the two-pass one-hot as nibbles (alphabets of at most 15 classes; expansion = k_expand_chunks<nibbles>, or
//                         k_expand_rows1<nibbles> for one-byte rows): 0 automatic (rows of 24 ... 31 bytes, id matrices beyond 128 MB, every one-byte row), 1 never, 2 whenever they apply
//   two_pass_slice_mb     the two-pass one-hot in SLICES of position rows whose id scratch stays cache-resident: 0 automatic (slices of <= 96 MB once the
//                         id matrix exceeds 128 MB), > 0 that many MB per slice (whatever the size), < 0 never
//   tokens_pb8_pair       k_tokens_pb8_fast: the last position tile of a matrix whose padlen % 64 is 1 ... 32 shared by two sequence tiles (0 automatic, 1 never)
//   expand_rows1          the LDS-free expansion k_expand_rows1 (one-byte elements, rows of 3 ... 15 bytes): 0 automatic, 1 never, 2 whenever it applies
//   gather_small          bsq_gather_packed_device: 0 one launch up to 4096 indices (k_gather_small), two beyond (k_gather_lengths2 + k_gather_place); 1 the three launches of rounds 2-5
//   host_pieces           list / host batch -> seq-first one-hot on the device: upload + encode in pieces (0 automatic: 4 pieces when the
//                         stream is idle and the batch large; 1 never; 2 ... 8 that many) -- bsq_host.cpp, piece_sequences()
//   fused_spins           fault injection (tests): polls of a fused launch's token wave before it gives up, poisons its chunk and reports
//                         (0: 2^18, about a second).  A wave that gives up is an ERROR of the call -- never a silent result
//   augment_k             attempts per lane and round of the augmentation kernel: 0 automatic (4), 1 (the round-2 form), 2
// The knobs are ONE plain struct, published as an immutable snapshot: a launcher reads it with a single atomic load
// (tuning()), never a name lookup under a mutex.  bsq_tuning_set() copies the current snapshot, changes one field and
// publishes the copy.
#define BSQ_KNOB_LIST(X)                                                                                                \
    X(nt_stores, 1) X(onehot_tb, 0) X(tile_order, 0) X(fill_mode, 0) X(onehot_path, 0) X(expand_pad, 0) X(tokenize_path, 0)   \
    X(fill_pad, 0) X(chunks_pad, 0) X(host_copy_threads, 0) X(tokenize_pad, 0) X(expand_slots, 0) X(tile_group, 0)             \
    X(bcl_path, 0) X(bcl_pad, 0) X(raw_mode, 0) X(workspace_cache, 0) X(tokens8, 0) X(tokens8_fast, 0) X(tokens8_lookup, 0)    \
    X(tokens8_pad, 0) X(pattern_wait, 0) X(tokenize_tb, 0) X(wide_index, 0) X(augment_k, 0) X(tokens_pb8, 0) X(augment_fused, 0) X(fused_spins, 0) X(expand_gate, 0) X(host_pieces, 0) X(gather_small, 0) X(expand_rows1, 0) X(raw_nibbles, 0) X(two_pass_slice_mb, 0) X(tokens_pb8_pair, 0)
struct Tuning {
#define BSQ_KNOB_FIELD(name, def) int32_t name = def;
    BSQ_KNOB_LIST(BSQ_KNOB_FIELD)
#undef BSQ_KNOB_FIELD
};
const Tuning &tuning();
int get_tuning(const char *name);             // by name (bsq_tuning_get); 0 for unknown names
bool set_tuning(const char *name, int value);  // false: unknown name
inline bool nontemporal_stores() { return tuning().nt_stores != 0; }

// Stream-ordered scratch: the buffer of the last few (device, stream) pairs is kept between calls (knob
// "workspace_cache" = 1: a hipMallocAsync / hipFreeAsync pair per call instead, as under graph capture).
bsq_status workspace_acquire(size_t nbytes, hipStream_t stream, void **ptr);
void workspace_release(void *ptr, hipStream_t stream);
void workspace_drop_cache();  // bsq_release_staging()
// Held by a caller from workspace_acquire() until its last launch that uses the scratch is enqueued: calls on one
// stream share the cached buffer, so their launches must not interleave.
std::mutex &workspace_mutex();

// bsq_tokens8.hip: the (B,P) int8 token matrix (register-table lookups, LDS rule tables).
bool tokens_bp8_applicable(const bsq_desc *d, int64_t B, int64_t P, const void *out);
int64_t tokens_bp8_chunks(int64_t B, int64_t P);                                              // 4-KiB chunks of the (B,P) int8 matrix
bool tokens_bp8_fast_form(const bsq_desc *d, int64_t B, int64_t P, bool aligned_out);         // k_tokens_bp8_fast (else k_tokens_bp8)
bool tokens_bp8_nowait_form(int64_t nchunks);                                                 // fused augmentation: no-wait form (else the flag form)
// raw = false: token VALUES (batch_tokenize); raw = true: ids with BSQ_NO_TOKEN (255) where a one-hot row is all zero
// fuse != nullptr: the BLOSUM62 augmentation of `fuse->chars` (the same buffer as `chars`) in the SAME launch when the fast form applies
// (*fused_taken = true); otherwise NOTHING is launched and the caller runs the two launches itself.
struct FusedAugRequest {
    uint8_t *chars;
    int32_t chain_len;
    double frac;
    uint64_t seed;
};
bsq_status launch_tokens_bp8(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets, int64_t B, int64_t P,
                             void *out, hipStream_t stream, bool raw = false, const uint8_t *mask = nullptr,
                             const FusedAugRequest *fuse = nullptr, bool *fused_taken = nullptr);
bsq_status augment_device_table(const void **table);  // bsq_augment.hip: the AugTable of the current device
uint32_t fused_failures();                              // bsq_tokens8.hip: token waves of fused launches that gave up waiting so far (sticky)
void fused_failures_clear();
// the (P,B) token matrix of any element type (pitch = elements between two position rows), no mask, 16-byte aligned rows
bool tokens_pb8_applicable(const bsq_desc *d, int64_t B, int64_t P, const void *out, int64_t pitch, bsq_dtype t = BSQ_I8);
// nib (raw only): ids as nibbles, two sequences per byte (15 = no one), a row = pitch / 2 bytes; at most 15 classes, pitch % 32 == 0
bsq_status launch_tokens_pb8(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets, int64_t B, int64_t P, void *out,
                             int64_t pitch, hipStream_t stream, bool raw = false, bsq_dtype t = BSQ_I8, bool nib = false,
                             int64_t tt0 = 0, int64_t ntt_count = 0);  // tt0, ntt_count: a SLICE of 64-position tiles (0 = all from tt0 on); `out` = its first row

// bsq_tokens8.hip: n <= 8 independent batches in one launch of the fast token kernel of their layout; *taken = false: nothing launched
bsq_status launch_tokens_multi(const bsq_desc *d, int32_t n, const bsq_batch *batches, int64_t P, bool batch_first, bsq_dtype t,
                               hipStream_t stream, bool *taken);

// bsq_tokens.hip: the channels-first (B, C, P) one-hot through the (B,P) chunk kernel's HOT form (the fallback of bsq_onehot_bcl_device for outputs
// below its two-pass threshold and for masked batches)
bsq_status launch_onehot_bcl_chunks(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets, const uint8_t *mask_or_null, int64_t B,
                                    int64_t P, bsq_dtype t, void *out, hipStream_t s);
// bsq_generic.hip: the element kernels as blocks of a wider destination (row_seqs sequences per position row; = B: the whole tensor / matrix)
bsq_status onehot_generic_block(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets, const uint8_t *mask_or_null, int64_t B,
                                int64_t P, bsq_dtype t, void *out, int64_t row_seqs, void *hip_stream);
bsq_status onehot_generic_bcl(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets, const uint8_t *mask_or_null, int64_t B, int64_t P,
                              bsq_dtype t, void *out, void *hip_stream);
bsq_status tokenize_generic_block(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets, int64_t B, int64_t P,
                                  int32_t batch_first, bsq_dtype t, void *out, int64_t row_seqs, void *hip_stream);

}  // namespace bsq_internal
