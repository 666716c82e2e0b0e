/*
 * bsq.h -- C ABI of libbsq_hip.so: the MI355X (gfx950) batch tokenizer / one-hot encoder.
 *
 * This is the drop-in boundary for the reference's hot path.  The reference (dnbaker/bioseq)
 * has no C ABI of its own -- its pybind11 lambdas call C++ templates directly
 * (/root/reference/src/tokenize.cpp:65-98) -- so each entry point below names the reference
 * function it replaces.  Plain pointers and sizes only; no torch / pybind / STL types.
 * Status-code returns, no exceptions cross the boundary, the caller owns every buffer.
 *
 * Batch representation ("packed batch", same CSR layout as the reference's FlatFile,
 * /root/reference/src/fxstats.cpp:33-64):
 *     chars   : uint8[total]   all sequences' bytes, concatenated
 *     offsets : int64[B + 1]   sequence i is chars[offsets[i] .. offsets[i+1])
 *     mask    : uint8[total] or NULL, one byte per input character, same offsets
 *
 * Measurement / diagnostic exports (tuning knobs, write-bandwidth yardsticks, self-tests) are declared in
 * bsq_diag.h, not here: they are not part of the drop-in surface.
 *
 * Output layouts (C-contiguous, bit-exact with the reference's numpy results):
 *     bsq_tokenize_* : (B, P) when batch_first else (P, B)       -- tokenize.h:420-425
 *     bsq_onehot_*   : (P, B, C), C = bsq_alphabet_size(desc)    -- tokenize.h:326-330
 */
#ifndef BSQ_H
#define BSQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BSQ_ABI_VERSION 7 /* 7 (round 6): bsq_tokenize_device_multi, bsq_augment_device_multi, bsq_augment_tokenize_device_multi, bsq_enable_peer_access, bsq_tokenize_kernel_name; nothing removed */

typedef int32_t bsq_status;
enum {
    BSQ_OK = 0,
    BSQ_ERR_INVALID_KEY = 1,  /* unknown alphabet key            (RuntimeError in Python, tokenize.h:74-79) */
    BSQ_ERR_INVALID_ARG = 2,  /* NULL pointer, negative size, padlen <= 0 (ValueError, tokenize.h:383)       */
    BSQ_ERR_DTYPE = 3,        /* unsupported dtype character     (ValueError, tokenize.cpp:80,97)            */
    BSQ_ERR_SEQ_TOO_LONG = 4, /* len + bos + eos > padlen        (reference aborts, tokenize.h:359-362,456)  */
    BSQ_ERR_NO_DEVICE = 5,    /* no HIP device visible: the product has NO CPU fallback                      */
    BSQ_ERR_HIP = 6,          /* a HIP runtime call failed; see bsq_last_error()                             */
    BSQ_ERR_ALLOC = 7,
    BSQ_ERR_FUSED_WAIT = 8    /* a fused augmentation + token launch gave up waiting inside the kernel: see bsq_fused_status()  */
};

/* Effective element types of the batch entry points.  The reference lower-cases the dtype
 * character before dispatch (tokenize.cpp:66,83), so 'B' is int8 and 'L'/'Q' are uint64. */
typedef enum { BSQ_I8 = 0, BSQ_I16 = 1, BSQ_I32 = 2, BSQ_U64 = 3, BSQ_F32 = 4, BSQ_F64 = 5 } bsq_dtype;

/* Where a buffer handed to a *_host entry point lives. */
typedef enum { BSQ_SPACE_HOST = 0, BSQ_SPACE_DEVICE = 1 } bsq_space;

/* Immutable tokenizer description == the state of the reference's `struct Tokenizer`
 * (tokenize.h:9-13): alphabet table + the three flags.  POD; copy freely. */
typedef struct bsq_desc {
    int8_t lut[256]; /* byte -> group id, -1 = unmapped (alphabet.h:32-61); bytes >= 0x80 are unmapped */
    int32_t nchars;  /* number of groups (alphabet.h:27)                                               */
    int32_t eos;     /* flags, in the reference's positional ctor order (key, eos, bos, padchar)       */
    int32_t bos;
    int32_t padchar;
} bsq_desc;

/* ---- library / errors ------------------------------------------------------------------- */
int32_t bsq_abi_version(void);
const char *bsq_strerror(bsq_status s);
/* Thread-local detail of the last failing call on this thread ("" if none). */
const char *bsq_last_error(void);
/* Number of visible HIP devices (0 if none / runtime unavailable). */
int32_t bsq_device_count(void);

/* ---- alphabets: replaces alph::CAMAP + TAlphabet::make_lut (alphabet.h:32-61,198-222) ---- */
int32_t bsq_num_keys(void);
const char *bsq_key_name(int32_t i);
/* key is matched case-insensitively (tokenize.h:73). */
bsq_status bsq_lut_get(const char *key, int8_t lut[256], int32_t *nchars);
/* Tokenizer(key, eos, bos, padchar) (tokenize.h:72-106, tokenize.cpp:23). */
bsq_status bsq_desc_init(bsq_desc *d, const char *key, int32_t eos, int32_t bos, int32_t padchar);
/* tokenize.h:21-33 */
int32_t bsq_bos_id(const bsq_desc *d);        /* -1 when bos is off  */
int32_t bsq_eos_id(const bsq_desc *d);        /* -1 when eos is off  */
int32_t bsq_pad_id(const bsq_desc *d);        /* returned even when padchar is off */
int32_t bsq_alphabet_size(const bsq_desc *d); /* C = nchars + eos + bos + padchar  */
/* dtype character dispatch of tokenize.cpp:65-98 (first character only, case-folded). */
bsq_status bsq_dtype_from_destchar(char c, bsq_dtype *out);
size_t bsq_dtype_size(bsq_dtype t);

/* ---- validation: the length check of tokenize.h:456-459 / :359-362, done BEFORE launch ---- */
/* offsets in host memory.  *first_bad = index of the first offending sequence or -1. */
bsq_status bsq_validate_lengths(const int64_t *offsets, int64_t B, int64_t P, int32_t bos, int32_t eos,
                                int64_t *first_bad);
/* offsets in device memory; runs a reduction kernel on `hip_stream` and synchronises it. */
bsq_status bsq_validate_lengths_device(const int64_t *offsets_dev, int64_t B, int64_t P, int32_t bos,
                                       int32_t eos, int64_t *first_bad, void *hip_stream);

/* The same plus the well-formedness of the offsets themselves: offsets[0] >= 0, non-decreasing, offsets[B] <= nchars
 * (the kernels bound their reads by offsets[B]).  BSQ_ERR_INVALID_ARG with *first_bad = the first offending entry, or
 * BSQ_ERR_SEQ_TOO_LONG with *first_bad = the first over-long sequence; malformed offsets are reported first. */
bsq_status bsq_validate_packed_device(const int64_t *offsets_dev, int64_t B, int64_t P, int32_t bos, int32_t eos,
                                      int64_t nchars, int64_t *first_bad, void *hip_stream);

/* ---- device entry points: every pointer is device memory; stream-ordered; never synchronise.
 * Over-long sequences are clamped inside the kernels (memory-safe); call a validate function
 * first if the reference's error behaviour is wanted.  hip_stream: a hipStream_t (NULL = default).
 *
 * bsq_tokenize_device replaces Tokenizer::transencode<T> (tokenize.h:381-485, `batch_tokenize`).
 * bsq_onehot_device   replaces Tokenizer::tokenize<T>(py::sequence,...) (tokenize.h:283-371,
 *                     `batch_onehot_encode`); every output element is written exactly once
 *                     (no memset pass). */
bsq_status bsq_tokenize_device(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets, int64_t B,
                               int64_t P, int32_t batch_first, bsq_dtype t, void *out, void *hip_stream);
bsq_status bsq_onehot_device(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets,
                             const uint8_t *mask_or_null, int64_t B, int64_t P, bsq_dtype t, void *out,
                             void *hip_stream);
/* Channels-first one-hot (B, C, P): out[(b*C + c)*P + t] = (token(b,t) == c) -- what the reference's conv
 * models get from rearrange('length batch emb -> batch emb length') (bioseq/loaders.py:74), written
 * directly.  Same semantics (mask, BOS/EOS/PAD, unmapped -> zero row) as bsq_onehot_device. */
bsq_status bsq_onehot_bcl_device(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets,
                                 const uint8_t *mask_or_null, int64_t B, int64_t P, bsq_dtype t, void *out,
                                 void *hip_stream);
/* The same one-hot written as a COLUMN BLOCK of a larger (P, row_seqs, C) tensor: `out` points at element (0, b0, 0) of it and
 * row t of this batch goes to out + t * row_seqs * C * sizeof(T).  What a rank of a sharded job needs to store its
 * sequences straight into the whole-batch tensor of another GPU (peer-mapped memory, sharding.store_shard_into_root), and what
 * a host batch that arrives in pieces is encoded with (staged batches, below).  A block of rows >= 16 bytes (or of one-byte
 * elements with rows of 3 ... 15 bytes) whose position rows are whole 4-KiB chunks (B * C * sizeof(T) and `out` multiples of
 * 4096), or any such block of 128 MB and more whatever its first sequence and the tensor's pitch, runs at the speed of the
 * whole-tensor stream: every position row of the block is cut at the 4-KiB boundaries of MEMORY, only its first and last
 * piece are partial.  Smaller unaligned blocks, masked small-row blocks go through the tiled kernel. */
bsq_status bsq_onehot_block_device(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets,
                                   const uint8_t *mask_or_null, int64_t B, int64_t P, bsq_dtype t, void *out, int64_t row_seqs,
                                   void *hip_stream);
/* batch_tokenize's DEFAULT layout -- the (P, B) token matrix -- as a column block of a wider (P, row_seqs) matrix: `out` points at
 * element (0, b0).  1-, 2- and 8-byte types of alphabets with ids < 251 at the speed of the whole matrix (any block width: widths
 * that are not a multiple of 16 bytes take the element-aligned form of 1- / 2-byte types); everything else through the generic
 * kernel.  (The (B, P) matrix needs no block form: rows [b0, b0 + n) are contiguous -- bsq_tokenize_device on a sub-batch.) */
bsq_status bsq_tokenize_block_device(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets, int64_t B, int64_t P,
                                     bsq_dtype t, void *out, int64_t row_seqs, void *hip_stream);
/* SEVERAL INDEPENDENT BATCHES IN ONE CALL (round 6).  The reference encodes one batch per call and its training loop issues the
 * calls back to back (bioseq/loaders.py:76-104; Tokenizer::transencode, tokenize.h:451-479, is one OpenMP region per batch); on the
 * GPU a 16-40-us token launch pays its own ramp-up and drain, and on one in-order stream the next batch cannot start under the tail
 * of this one.  bsq_tokenize_device_multi encodes n packed batches of ONE tokenizer, padlen, layout and element type -- each with its
 * own characters, offsets and output matrix ((B_i, P) or (P, B_i), contiguous) -- with results identical to n calls of
 * bsq_tokenize_device on the same stream, in ceil(n / 8) launches when every batch qualifies for the fast kernel of its layout
 * (int8 (B,P) with padlen % 16 == 0 and >= 128; 1- / 2-byte (P,B) with 64-byte aligned rows; ids < 251), and as n launches otherwise.
 * Batches with B == 0 are skipped.  n < 0 or a null table: BSQ_ERR_INVALID_ARG.  The batches' buffers must not overlap one another
 * (they are independent: nothing orders one batch's stores against another's loads inside the launch). */
typedef struct bsq_batch {
    const uint8_t *chars;   /* device: packed characters of this batch */
    const int64_t *offsets; /* device: B + 1 offsets into chars */
    int64_t B;              /* sequences */
    void *out;              /* device: B * P elements, (B, P) or (P, B) */
} bsq_batch;
bsq_status bsq_tokenize_device_multi(const bsq_desc *d, int32_t n, const bsq_batch *batches, int64_t P, int32_t batch_first,
                                     bsq_dtype t, void *hip_stream);
/* Name of the kernel(s) bsq_onehot_device would launch for this shape (profiling / bench labels). */
const char *bsq_onehot_kernel_name(const bsq_desc *d, int64_t B, int64_t P, bsq_dtype t);
/* The same for bsq_tokenize_device (augment = 0) and bsq_augment_tokenize_device (augment = its chain_len > 0), for a 16-byte aligned
 * contiguous output: which token kernel, and for the fused augmentation which of its two one-launch forms. */
const char *bsq_tokenize_kernel_name(const bsq_desc *d, int64_t B, int64_t P, int32_t batch_first, bsq_dtype t, int32_t augment);
/* Same results through the simple one-thread-per-element kernels (any shape/alignment/alphabet).
 * Used as the in-library cross-check of the tiled kernels and as their fallback. */
bsq_status bsq_tokenize_device_generic(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets,
                                       int64_t B, int64_t P, int32_t batch_first, bsq_dtype t, void *out,
                                       void *hip_stream);
bsq_status bsq_onehot_device_generic(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets,
                                     const uint8_t *mask_or_null, int64_t B, int64_t P, bsq_dtype t, void *out,
                                     void *hip_stream);
/* The two passes of the large-output one-hot path on their own (distributed assembly: ship the small token
 * matrices over xGMI and expand at the destination -- 1/(C*sizeof(T)) of the one-hot's bytes, e.g. 1/80 at cfg3):
 * bsq_raw_tokens_device       -> tokens[t * pitch + b] = id at position t of sequence b (tokenize.h:342-369
 *                                semantics incl. BOS/EOS/PAD/mask), BSQ_NO_TOKEN where the one-hot row is all zero;
 *                                pitch >= B (a multiple of 16 and a 16-byte aligned base make the stores vectorised;
 *                                columns B..pitch-1 of every row are scratch and may be overwritten);
 * bsq_onehot_from_raw_tokens_device -> out[(t*B + b)*C + c] = (tokens[t*pitch + b] == c), every element written once.
 * Together they equal bsq_onehot_device.  Limits: ids < 251 (not BYTES), padlen <= 2^22. */
#define BSQ_NO_TOKEN 255
bsq_status bsq_raw_tokens_device(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets,
                                 const uint8_t *mask_or_null, int64_t B, int64_t P, uint8_t *tokens, int64_t pitch,
                                 void *hip_stream);
bsq_status bsq_onehot_from_raw_tokens_device(const uint8_t *tokens, int64_t pitch, int64_t B, int64_t P, int32_t C,
                                             bsq_dtype t, void *out, void *hip_stream);

/* ---- decode on the device: replaces Tokenizer::decode_tokens (tokenize.h:131-183) for token matrices that live in
 * HBM (README.md:48: "if you have logits, use an argmax to convert to tokens for decoding").  The tokens never travel
 * to the host, only the decoded text does.  A token decodes to the first byte of its alphabet group or to one of the
 * five-byte pieces <BOS> / <EOS> / <PAD>; rows are decoded independently (1-D input: nrows = 1).
 *   tokens: device pointer, element size `itemsize` (1, 2, 4 or 8: unsigned loads of tokenize.h:107-124), element
 *           (r, c) at byte offset r * row_stride + c * col_stride (any strides that are multiples of itemsize).
 * bsq_decode_sizes_device: row_offsets (device, nrows + 1 int64) <- exclusive prefix sum of the decoded row lengths,
 *   *total <- bytes of all rows; synchronises the stream.  A token outside the tokenizer's table gives
 *   BSQ_ERR_INVALID_ARG with *first_bad = r * ncols + c of the first one (the reference throws "Unexpected/invalid
 *   token"), else *first_bad = -1.
 * bsq_decode_write_device: row r's text into out_chars[row_offsets[r] .. row_offsets[r + 1]); stream-ordered.
 * bsq_argmax_tokens_device: tokens[r] = argmax_c logits[r * row_stride + c] (first maximum, like torch.argmax),
 *   r < n, channels contiguous; logit_kind BSQ_LOGITS_*; tokens uint8 (token_itemsize 1, C <= 256) or int32 (4). */
enum { BSQ_LOGITS_F32 = 0, BSQ_LOGITS_F64 = 1, BSQ_LOGITS_F16 = 2, BSQ_LOGITS_BF16 = 3 };
bsq_status bsq_decode_sizes_device(const bsq_desc *d, const void *tokens, int32_t itemsize, int64_t nrows, int64_t ncols,
                                   int64_t row_stride, int64_t col_stride, int64_t *row_offsets, int64_t *total,
                                   int64_t *first_bad, void *hip_stream);
bsq_status bsq_decode_write_device(const bsq_desc *d, const void *tokens, int32_t itemsize, int64_t nrows, int64_t ncols,
                                   int64_t row_stride, int64_t col_stride, const int64_t *row_offsets, uint8_t *out_chars,
                                   void *hip_stream);
bsq_status bsq_argmax_tokens_device(const void *logits, int32_t logit_kind, int64_t n, int32_t C, int64_t row_stride,
                                    void *tokens, int32_t token_itemsize, void *hip_stream);

/* ---- BLOSUM62 augmentation (the pre-step of BASELINE config 5): replaces bioseq/blosum.py:36-87.
 * bsq_blosum62_normrows: the 21x20 float64 transition table `normrows` (rows ARNDCQEGHILKMFPSTWYV+X,
 * columns ARNDCQEGHILKMFPSTWYV), bit-identical to the reference's numpy result.
 * bsq_augment_device: in place on a packed batch in device memory; every sequence is mutated with
 * probability `frac` (>= 1: always) by `chain_len` BLOSUM62-weighted point substitutions (new != old),
 * unknown residues use the X row.  Deterministic in (seed, sequence index); stream-ordered. */
bsq_status bsq_blosum62_normrows(double *out21x20);
bsq_status bsq_augment_device(uint8_t *chars, const int64_t *offsets, int64_t B, int32_t chain_len, double frac,
                              uint64_t seed, void *hip_stream);
/* bsq_augment_tokenize_device: bsq_augment_device followed by bsq_tokenize_device on the same packed batch -- what the reference's
 * loaders do per item (bioseq/loaders.py:83-84, :102-103: augment_seq, then batch_tokenize) -- with exactly their results
 * (`chars` mutated in place, `out` the token matrix of the mutated batch).  For (B,P) int8 matrices that the fast token kernel takes
 * (padlen % 16 == 0, chains of <= 4 mutations) it is ONE launch: the augmentation's workgroups come first and publish every mutation
 * in a side list; the token workgroups encode at the same time and patch the mutated positions once their rows' augmentation is
 * done.  Every other shape, and a stream under graph capture, runs the two launches.
 *
 * The one-launch form waits INSIDE the kernel, bounded (about a second).  A token wave whose wait expires -- never observed; it would
 * take a dispatcher that starts workgroups out of order -- overwrites its 4 KiB of `out` with 0xFF bytes (no token matrix contains
 * them) and counts itself in host-visible memory.  That count is STICKY: while it is non-zero bsq_augment_tokenize_device returns
 * BSQ_ERR_FUSED_WAIT at entry, and bsq_fused_status reports it.  Callers check bsq_fused_status after synchronising the stream; there
 * is no state in which wrong tokens coexist with BSQ_OK from that check.  What the Python layer checks by itself: every
 * augment_tokenize_packed call at its entry (the sticky count: an earlier launch's failure), FlatFileDataset.batches() before it
 * yields each batch (completed launches) and, synchronised, once the epoch's last batch has been handed out -- RuntimeError in each
 * case; any other consumer calls blosum.check_fused(synchronize=True) before it trusts a batch.
 * bsq_fused_status: *failures (nullable) <- token waves that gave up since the last clear; BSQ_OK iff 0, else BSQ_ERR_FUSED_WAIT.
 * Reads host memory only: no synchronisation, callable at any time; it covers the launches that have COMPLETED.
 * bsq_fused_status_clear: forget the count (after the caller has discarded the poisoned outputs). */
bsq_status bsq_augment_tokenize_device(const bsq_desc *d, uint8_t *chars, const int64_t *offsets, int64_t B, int64_t P,
                                       int32_t batch_first, bsq_dtype t, void *out, int32_t chain_len, double frac,
                                       uint64_t seed, void *hip_stream);
bsq_status bsq_fused_status(uint32_t *failures);
void bsq_fused_status_clear(void);
/* The same for n independent batches (`bsq_batch`, below; `chars` is mutated in place, seeds[i] is batch i's seed): results identical to n
 * calls of bsq_augment_device / bsq_augment_tokenize_device with those seeds.  The augmentations of up to eight batches are ONE launch
 * (a batch's augmentation is a short, latency-bound generation of waves: eight cost about as much as one), their token matrices one more
 * (bsq_tokenize_device_multi) -- no wait inside a kernel, every character read once: BASELINE config 5 with augmentation on fresh batches,
 * four per call, runs at 0.6 of the HBM roof where one batch per call reaches 0.45.  This is how a training loop
 * that has its next batches at hand (bioseq/loaders.py:76-104) should call the path. */
bsq_status bsq_augment_device_multi(int32_t n, const bsq_batch *batches, int32_t chain_len, double frac, const uint64_t *seeds,
                                    void *hip_stream);
bsq_status bsq_augment_tokenize_device_multi(const bsq_desc *d, int32_t n, const bsq_batch *batches, int64_t P, int32_t batch_first,
                                             bsq_dtype t, int32_t chain_len, double frac, const uint64_t *seeds, void *hip_stream);

/* ---- index-list batches from a packed store resident in HBM: replaces the per-item fetch of FlatFileDataset.__getitem__
 * (bioseq/loaders.py:76-104: ff.access(i) on the host for every sample) under a shuffling sampler.  Rebuilds the packed
 * batch of sequences index[0 .. n) of the store (chars, offsets: n_store sequences) on the device:
 *     out_offsets[0] = 0, out_offsets[i + 1] - out_offsets[i] = length of sequence index[i];
 *     out_chars[out_offsets[i] .. out_offsets[i + 1]) = its characters
 * -- ready for bsq_tokenize_device / bsq_onehot_device / bsq_onehot_bcl_device / bsq_augment_device.  Indices may
 * repeat, in any order; empty sequences are fine.  Stream-ordered, never synchronises: errors are left in *status_dev
 * (device int64): -1 = ok, i in [0, n) = index[i] was out of range (it contributes an empty sequence), n + i = output
 * sequence i did not fit into out_capacity bytes (the batch is cut there, nothing is written past the buffer).
 * out_capacity = n * (longest sequence of the store) always suffices.  out_chars may be NULL to get the offsets only;
 * status_dev may be NULL when the caller vouches for its indices and capacity (nothing is reported; a bad index still reads
 * nothing, an overflow is still cut).  Lists of up to 4096 indices -- a training step's batch -- take ONE launch. */
bsq_status bsq_gather_packed_device(const uint8_t *chars, const int64_t *offsets, int64_t n_store, const int64_t *index,
                                    int64_t n, uint8_t *out_chars, int64_t out_capacity, int64_t *out_offsets,
                                    int64_t *status_dev, void *hip_stream);

/* ---- FASTA / FASTQ (plain or gzip) -> FlatFile on the host: replaces FlatFile::make (fxstats.cpp:33-64) and getlens /
 * getstats (:12-23, :202-219).  Same record grammar as the reference's kseq loop (bsq_fastx.cpp lists it), but streaming:
 * only the offsets stay in memory.  File format: uint64 nseqs | uint64 offsets[nseqs + 1] | sequence bytes -- the packed
 * batch itself.  bsq_fastx_lengths: lens[0 .. min(n, capacity)) <- sequence length per record, *nrecords <- n (call with
 * capacity 0 to count).  Host-only: no device is needed. */
bsq_status bsq_fastx_to_flatfile(const char *inpath, const char *outpath, int64_t *nseqs, int64_t *max_seq_len);
bsq_status bsq_fastx_lengths(const char *path, uint64_t *lens, int64_t capacity, int64_t *nrecords);

/* ---- host entry points: packed batch in HOST memory (pageable or pinned).  The library stages
 * it through its own pinned + device buffers on the current HIP device, runs the device entry
 * point on `hip_stream`, and leaves the result in `out`:
 *   out_space == BSQ_SPACE_DEVICE : out is device memory, the call returns after enqueueing;
 *   out_space == BSQ_SPACE_HOST   : out is host memory, the call returns after the D2H copy.
 * Lengths are validated first (BSQ_ERR_SEQ_TOO_LONG, *first_bad set, nothing launched). */
bsq_status bsq_tokenize_host(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets, int64_t B,
                             int64_t P, int32_t batch_first, bsq_dtype t, void *out, bsq_space out_space,
                             void *hip_stream, int64_t *first_bad);
bsq_status bsq_onehot_host(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets,
                           const uint8_t *mask_or_null, int64_t B, int64_t P, bsq_dtype t, void *out,
                           bsq_space out_space, void *hip_stream, int64_t *first_bad);

bsq_status bsq_onehot_bcl_host(const bsq_desc *d, const uint8_t *chars, const int64_t *offsets,
                               const uint8_t *mask_or_null, int64_t B, int64_t P, bsq_dtype t, void *out,
                               bsq_space out_space, void *hip_stream, int64_t *first_bad);

/* ---- staged batches: for a caller that PRODUCES a host batch piece by piece (the pybind11 layer scanning and copying Python
 * objects) and wants piece j on the bus and under the encode kernels while it produces piece j + 1.
 *   bsq_stage_begin   takes the next pinned + device staging slot of the current device (waits until the batch that used it has
 *                     left the GPU) sized for max_seqs sequences / max_chars characters, and returns the PINNED host buffers to
 *                     write: offsets[max_seqs + 1] (offsets[0] = 0 is set), chars[max_chars], mask[max_chars] (with_mask).
 *   bsq_stage_upload  sequences [first, last) are complete -- offsets[first + 1 .. last] and their characters written; pieces
 *                     follow one another from 0.  Enqueues their copy on the library's copy stream, makes `hip_stream` wait for
 *                     it, and returns DEVICE pointers for a *_device call on hip_stream over those sequences: d_offsets points
 *                     at the entry of sequence `first` (values are offsets into d_chars / d_mask, the bases of the whole batch).
 *   bsq_stage_end     always pairs with a successful begin: the slot may be reused once the work enqueued on hip_stream so far
 *                     has run.  Other bsq_*_host calls of the process wait between begin and end (one staging area per device).
 *   bsq_stage_piece_hint  sequences per piece the library recommends for a batch of B sequences / ~nchars characters whose
 *                     result blocks have block_row_bytes-byte rows at `out` (0: the blocks are contiguous), 0 = one piece, or
 *                     -1 = the knob asks for the whole-batch path (host_pieces = 1).  *head_seqs: sequences the FIRST piece holds
 *                     in front of that (pieces [0, head + n), [head + n, head + 2n), ...): column blocks run fastest when they
 *                     start where a 4-KiB chunk of the result starts, and `out` is rarely aligned that far -- encode the head
 *                     with a call of its own (head_seqs may be NULL: pieces [0, n), [n, 2n), ..., each cut at the chunk
 *                     boundaries of memory inside bsq_onehot_block_device).  Knob "host_pieces"; automatic = pieces of ~8 MB when the batch is large and hip_stream is idle (a busy
 *                     stream means the caller is not waiting for this batch: one upload costs the host less than several).
 * What it buys (list of 65 536 bytes objects, 35 MB -> f32 one-hot on the device, synchronous): 2.1 ms as one pack + one
 * upload + one encode, 1.5 ms with the encode and the pack of the pieces under the uploads (profiles/r04/host_pieces_lab.txt).
 *
 * A result that has to end up in HOST memory (the reference's default return is a numpy array):
 *   bsq_stage_result  a device scratch of nbytes for the pieces' results + a PINNED host area of the same size (both owned by the
 *                     staging area, valid until bsq_stage_end); encode piece j to d_result + offset_j with any *_device entry;
 *   bsq_stage_fetch   enqueue the copy of [offset, offset + nbytes) of the device scratch to the same offsets of the pinned area
 *                     on hip_stream (behind the encode of that piece; the upload of the next piece runs the other way meanwhile);
 *                     *ticket (may be NULL) names this fetch for bsq_stage_wait (-1: none left, wait for everything);
 *   bsq_stage_wait    block until fetch `ticket` has landed (ticket < 0: until everything enqueued on hip_stream has happened).
 * The caller then copies the pinned bytes where it wants them (the pybind layer: into the numpy array, with its worker pool).
 * list of 65 536 items -> numpy int8 (P, B) tokens, the reference's literal default call: 3.4 -> 2.3 ms (profiles/r04/default_call_lab.txt). */
typedef struct bsq_stage bsq_stage;
bsq_status bsq_stage_begin(int64_t max_seqs, size_t max_chars, int32_t with_mask, void *hip_stream, bsq_stage **stage,
                           int64_t **offsets, uint8_t **chars, uint8_t **mask);
bsq_status bsq_stage_upload(bsq_stage *stage, int64_t first, int64_t last, const int64_t **d_offsets, const uint8_t **d_chars,
                            const uint8_t **d_mask);
bsq_status bsq_stage_end(bsq_stage *stage);
bsq_status bsq_stage_result(bsq_stage *stage, size_t nbytes, void **d_result, void **h_result);
bsq_status bsq_stage_fetch(bsq_stage *stage, size_t offset, size_t nbytes, int32_t *ticket);
bsq_status bsq_stage_wait(bsq_stage *stage, int32_t ticket);
int64_t bsq_stage_piece_hint(int64_t B, size_t nchars, size_t block_row_bytes, const void *out, void *hip_stream, int64_t *head_seqs);

/* One process, several devices ("host packs once; GPU g receives its slice" -- the reference's only multi-GPU consumer is a
 * single-process nn.DataParallel, training/cnnpretrain.py:85-94): lets kernels launched on `device` store into memory of `peer`
 * (hipDeviceEnablePeerAccess; already enabled is not an error), so that the block entry points above can write a device's shard straight
 * into the whole-batch tensor that lives on another device.  device == peer: nothing to do. */
bsq_status bsq_enable_peer_access(int32_t device, int32_t peer);

/* Pinned host scratch for callers that pack Python objects themselves (the pybind11 layer): returns a buffer of
 * at least nbytes; pack offsets | chars | mask into it and hand those pointers to the next bsq_*_host call.
 * Three buffers take turns (the call waits until the batch packed three calls ago has left the GPU), so packing
 * batch n + 1 overlaps the copy + encode of batches n and n - 1; a buffer stays valid until the call after next. */
void *bsq_pinned_scratch(size_t nbytes);
/* Free every cached staging buffer of the calling process (tests, shutdown). */
void bsq_release_staging(void);

#ifdef __cplusplus
}
#endif
#endif /* BSQ_H */
