// BLOSUM62 point-mutation augmentation on the device.
//
// Replaces the per-sequence pure-Python chain of /root/reference/bioseq/blosum.py:63-87
// (`augment_seq`) and its table construction (:36-48 `normrows`): callers apply it to the byte strings
// right before batch_tokenize / batch_onehot_encode (bioseq/loaders.py:83,102;
// training/cnnpretrain.py:115-117).  Here it runs in place on the packed batch already in HBM.
//
// Distribution per sequence (the reference's): with probability `frac` mutate the sequence;
// repeat chain_len times { repeat { idx = uniform position; new = draw from normrows[row(seq[idx])] }
// until new != seq[idx]; seq[idx] = new }.  Unknown residues (anything outside the 20 letters, incl.
// lower case) use the 'X' row, as probdict.get(c, default_transitions) does.  The kernel samples the same
// joint distribution in two steps (accept the position with probability 1 - row[old], then draw the new
// residue from the row without its own entry): see k_augment.
// The random STREAM is our own (counter-based splitmix64 keyed by seed / sequence / mutation /
// attempt; the reference uses a module-global numpy PCG64 that also depends on import order), so
// parity is defined on the table (bit-exact), the invariants and the substitution statistics.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <type_traits>

#include "bsq.h"
#include "bsq_internal.h"

namespace {

constexpr int kRows = 21, kCols = 20;
constexpr char kLetters[] = "ARNDCQEGHILKMFPSTWYV";  // column order; row order is the same + 'X'

// BLOSUM62 scores, rows ARNDCQEGHILKMFPSTWYV + X, columns ARNDCQEGHILKMFPSTWYV (public NCBI matrix).
constexpr int8_t kBlosum62[kRows][kCols] = {
    /*A*/ {4, -1, -2, -2, 0, -1, -1, 0, -2, -1, -1, -1, -1, -2, -1, 1, 0, -3, -2, 0},
    /*R*/ {-1, 5, 0, -2, -3, 1, 0, -2, 0, -3, -2, 2, -1, -3, -2, -1, -1, -3, -2, -3},
    /*N*/ {-2, 0, 6, 1, -3, 0, 0, 0, 1, -3, -3, 0, -2, -3, -2, 1, 0, -4, -2, -3},
    /*D*/ {-2, -2, 1, 6, -3, 0, 2, -1, -1, -3, -4, -1, -3, -3, -1, 0, -1, -4, -3, -3},
    /*C*/ {0, -3, -3, -3, 9, -3, -4, -3, -3, -1, -1, -3, -1, -2, -3, -1, -1, -2, -2, -1},
    /*Q*/ {-1, 1, 0, 0, -3, 5, 2, -2, 0, -3, -2, 1, 0, -3, -1, 0, -1, -2, -1, -2},
    /*E*/ {-1, 0, 0, 2, -4, 2, 5, -2, 0, -3, -3, 1, -2, -3, -1, 0, -1, -3, -2, -2},
    /*G*/ {0, -2, 0, -1, -3, -2, -2, 6, -2, -4, -4, -2, -3, -3, -2, 0, -2, -2, -3, -3},
    /*H*/ {-2, 0, 1, -1, -3, 0, 0, -2, 8, -3, -3, -1, -2, -1, -2, -1, -2, -2, 2, -3},
    /*I*/ {-1, -3, -3, -3, -1, -3, -3, -4, -3, 4, 2, -3, 1, 0, -3, -2, -1, -3, -1, 3},
    /*L*/ {-1, -2, -3, -4, -1, -2, -3, -4, -3, 2, 4, -2, 2, 0, -3, -2, -1, -2, -1, 1},
    /*K*/ {-1, 2, 0, -1, -3, 1, 1, -2, -1, -3, -2, 5, -1, -3, -1, 0, -1, -3, -2, -2},
    /*M*/ {-1, -1, -2, -3, -1, 0, -2, -3, -2, 1, 2, -1, 5, 0, -2, -1, -1, -1, -1, 1},
    /*F*/ {-2, -3, -3, -3, -2, -3, -3, -3, -1, 0, 0, -3, 0, 6, -4, -2, -2, 1, 3, -1},
    /*P*/ {-1, -2, -2, -1, -3, -1, -1, -2, -2, -3, -3, -1, -2, -4, 7, -1, -1, -4, -3, -2},
    /*S*/ {1, -1, 1, 0, -1, 0, 0, 0, -1, -2, -2, 0, -1, -2, -1, 4, 1, -3, -2, -2},
    /*T*/ {0, -1, 0, -1, -1, -1, -1, -2, -2, -1, -1, -1, -1, -2, -1, 1, 5, -2, -2, 0},
    /*W*/ {-3, -3, -4, -4, -2, -2, -3, -2, -2, -3, -2, -3, -1, 1, -4, -3, -2, 11, 2, -3},
    /*Y*/ {-2, -2, -2, -3, -2, -1, -2, -3, 2, -1, -1, -2, -1, 3, -3, -2, -2, 2, 7, -1},
    /*V*/ {0, -3, -3, -3, -1, -2, -2, -3, -3, 3, 1, -2, 1, -1, -2, -2, 0, -3, -1, 4},
    /*X*/ {0, -1, -1, -1, -2, -1, -1, -1, -1, -1, -1, -1, -1, -1, -2, 0, 0, -2, -1, -1},
};

// normrows[a][:] = 2^score / sum(2^score)  (blosum.py:41-45).  Every 2^score is a power of two in
// [2^-4, 2^11], so the row sum is exact in double whatever the summation order, and the quotient is one
// correctly rounded IEEE division: bit-identical to numpy's result.
void make_normrows(double out[kRows * kCols]) {
    for (int r = 0; r < kRows; ++r) {
        double sum = 0.0;
        for (int c = 0; c < kCols; ++c) sum += std::ldexp(1.0, kBlosum62[r][c]);
        for (int c = 0; c < kCols; ++c) out[r * kCols + c] = std::ldexp(1.0, kBlosum62[r][c]) / sum;
    }
}

struct AugTable {
    double cdf[kRows][kCols];  // inclusive prefix sums of normrows (left to right)
    double self[kRows];        // normrows[r][r]: probability that a draw from row r repeats the residue (row X: 0)
    uint8_t row_of[256];       // byte -> row (20 = 'X' row for everything unknown)
    uint8_t letter[kCols];
    uint32_t accept_le[kRows];  // a position whose residue is of row r is accepted iff lo32 <= accept_le[r]: the integer form of
                                // `double(lo32) * 2^-32 < 1.0 - self[r]` (both sides exact doubles: the same truth value for every lo32)
    uint32_t pad_[4];
};

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// i-th 64-bit word of the stream keyed by (seed, sequence): mix64(h0 + 0xD1342543DE82EF95 * (i + 1)) with
// h0 = mix64(seed + 0x9E3779B97F4A7C15 * (seq + 1)).  Host twin: tests/test_augment.py.
#ifdef BSQ_LABS
__device__ __forceinline__ uint64_t rnd(uint64_t seed, uint64_t seq, uint64_t i) {
    return mix64(mix64(seed + 0x9E3779B97F4A7C15ull * (seq + 1)) + 0xD1342543DE82EF95ull * (i + 1));
}
#endif
__device__ __forceinline__ double unit(uint64_t x) { return static_cast<double>(x >> 11) * 0x1.0p-53; }
// floor(r * len / 2^64) for len < 2^32: two 32 x 32 multiplies instead of the four of __umul64hi (quarter-rate instructions)
__device__ __forceinline__ uint64_t mulhi_64x32(uint64_t r, uint32_t len) {
    const uint64_t lo = static_cast<uint64_t>(static_cast<uint32_t>(r)) * len;
    return (static_cast<uint64_t>(static_cast<uint32_t>(r >> 32)) * len + (lo >> 32)) >> 32;
}

constexpr int kMaxAttempts = 1 << 14;  // the reference's `while inchar == outchar` is unbounded (an all-W sequence accepts with p = 0.006 per try)

#ifdef BSQ_LABS  // round-1 form (knob augment_mode 1): kept for A/B runs in diagnostic builds only
// One mutation = the reference's loop `repeat { idx = choice(L); new = choice(letters, p = row(seq[idx])) } until
// new != seq[idx]` (bioseq/blosum.py:63-87), sampled in two steps with the same joint distribution: a position is
// ACCEPTED with probability 1 - row(old)[old] (what the reference's rejection amounts to), and only then is the
// new residue drawn -- from row(old) without its own entry.  An attempt costs one random word, one character
// gather and one table lookup instead of two words and a 20-step search; the gathers of kBatch attempts are
// issued together (attempts are still evaluated strictly in counter order, so results do not depend on kBatch).
// Random words of thread b: 0 = augment_frac decision; then one per attempt (high bits: position, low 32 bits:
// acceptance) and one per accepted mutation (the new residue).
__global__ __launch_bounds__(256) void k_augment(uint8_t *chars, const int64_t *offsets, int64_t B, int32_t chain_len,
                                                 double frac, uint64_t seed, const AugTable *tab) {
    __shared__ AugTable s_tab;
    for (int i = threadIdx.x; i < int(sizeof(AugTable) / 4); i += 256)
        reinterpret_cast<uint32_t *>(&s_tab)[i] = reinterpret_cast<const uint32_t *>(tab)[i];
    __syncthreads();
    const int64_t b = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (b >= B) return;
    const int64_t start = offsets[b];
    const int64_t L = offsets[b + 1] - start;
    if (L <= 0) return;
    if (frac < 1.0 && !(unit(rnd(seed, b, 0)) < frac)) return;  // word 0 decides whether b is augmented
    uint64_t ctr = 1;
    // One round of NB attempts: gathers first, then evaluation in counter order.  Returns true once a mutation is made.
    auto round = [&](auto nb_tag) -> bool {
        constexpr int NB = decltype(nb_tag)::value;
        int64_t idx[NB];
        uint32_t lo[NB];
        uint8_t old[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const uint64_t r = rnd(seed, b, ctr + j);
            idx[j] = static_cast<int64_t>(__umul64hi(r, static_cast<uint64_t>(L)));  // uniform in [0, L)
            lo[j] = static_cast<uint32_t>(r);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) old[j] = chars[start + idx[j]];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            ctr += 1;
            const int row = s_tab.row_of[old[j]];
            const double pself = s_tab.self[row];
            if (static_cast<double>(lo[j]) * 0x1.0p-32 < 1.0 - pself) {  // position accepted
                const double *cdf = s_tab.cdf[row];
                const double u = unit(rnd(seed, b, ctr)) * (cdf[kCols - 1] - pself);
                ctr += 1;
                int pick = -1;  // first k != row with u < cdf[k] minus the removed diagonal mass; else the last such k
                for (int k = 0; k < kCols; ++k) {
                    if (k == row) continue;
                    pick = k;
                    if (u < cdf[k] - (k > row ? pself : 0.0)) break;
                }
                chars[start + idx[j]] = s_tab.letter[pick];
                return true;
            }
        }
        return false;
    };
    // The kernel ends with its unluckiest thread (~30 attempts among 10^5 threads at 30 % acceptance), and every
    // round is a dependent memory round trip: 4 attempts in the first round, 16 in the later ones.
    for (int32_t m = 0; m < chain_len; ++m) {
        bool done = round(std::integral_constant<int, 4>{});
        for (int a0 = 4; a0 < kMaxAttempts && !done; a0 += 16) done = round(std::integral_constant<int, 16>{});
    }
}
#endif  // BSQ_LABS

// Attempt-parallel form of the same algorithm and the SAME random stream (results are identical to k_augment; the
// numpy twin in tests/test_augment.py is the judge of both).  k_augment gives every sequence one lane, so a wave runs
// until its unluckiest lane is accepted and ~93 % of its vector work is spent on lanes that are already done or were
// never selected (frac = 0.5), all of it 64-bit multiplies of the counter RNG: 27 us on cfg5.  Here a wave owns 64
// sequences (state in LDS, one home lane each) and spends its 64 lanes on A ATTEMPTS x 64/A pending sequences per
// step, A = 64 / (pending sequences rounded up to a power of two): lane (g, a) evaluates attempt ctr + a of the g-th
// pending sequence (one random word, one gathered character, one table lookup), a group ballot finds the first
// accepted attempt in counter order, that lane draws the new residue and writes it.  32 pending sequences take 2
// attempts each, the ~16 left 4 each, the ~4 left 16 each: three or four dependent memory round trips per wave.
constexpr int kSeqPerWave = 64;
template <int K>
__global__ __launch_bounds__(256) void k_augment_groups(uint8_t *chars, const int64_t *offsets, int64_t B, int32_t chain_len,
                                                        double frac, uint64_t seed, const AugTable *tab) {
    __shared__ __align__(16) AugTable s_tab;
    __shared__ int64_t s_start[4][kSeqPerWave], s_len[4][kSeqPerWave];
    __shared__ uint64_t s_h0[4][kSeqPerWave];
    __shared__ uint32_t s_ctr[4][kSeqPerWave];
    __shared__ int32_t s_rem[4][kSeqPerWave], s_tries[4][kSeqPerWave];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t b = (static_cast<int64_t>(blockIdx.x) * 4 + wave) * kSeqPerWave + lane;
    // the spans first: their loads are in flight while the table is staged (round 3: the kernel is latency-bound --
    // 59 % of its wave cycles are waits, profiles/r03/augment_groups_pmc.txt -- and this was one dependent round trip more)
    int64_t start = 0, L = 0;
    if (b < B) {
        start = offsets[b];
        L = offsets[b + 1] - start;
    }
    static_assert(sizeof(AugTable) % 16 == 0, "staged as 16-byte pieces");
    for (int i = threadIdx.x; i < int(sizeof(AugTable) / 16); i += 256)
        reinterpret_cast<uint4 *>(&s_tab)[i] = reinterpret_cast<const uint4 *>(tab)[i];
    __syncthreads();
    {   // home lanes: which sequences are augmented at all (word 0 of their stream), their keys
        int32_t rem = 0;
        uint64_t h0 = 0;
        if (b < B) {
            h0 = mix64(seed + 0x9E3779B97F4A7C15ull * (static_cast<uint64_t>(b) + 1));
            const bool pick = L > 0 && (!(frac < 1.0) || unit(mix64(h0 + 0xD1342543DE82EF95ull)) < frac);
            rem = pick ? chain_len : 0;
        }
        // FIRST ROUND, in the home lane itself (round 3): the K attempts with counters 1 .. K of the first mutation, their gathers
        // in flight together, straight out of registers.  An attempt is accepted with probability 1 - p_self (~0.75 on the
        // average protein), so K = 4 settle 99.6 % of the mutations here and the typical wave never enters the group machinery
        // below -- no state round trip through LDS, no second pass over the loop (cycle-counter timeline of the kernel:
        // profiles/r03/augment_timeline.txt).  Attempts are consumed in counter order as everywhere: same results.
        uint32_t ctr = 1;
        int32_t tries = 0;
        const bool long_len0 = __builtin_amdgcn_ballot_w64((static_cast<uint64_t>(L) >> 32) != 0) != 0;
        if (rem > 0 && !long_len0) {
            uint64_t r[K];
            int64_t idx[K];
            uint8_t ch[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                r[k] = mix64(h0 + 0xD1342543DE82EF95ull * (static_cast<uint64_t>(ctr + k) + 1));
                idx[k] = static_cast<int64_t>(mulhi_64x32(r[k], static_cast<uint32_t>(L)));
            }
#pragma unroll
            for (int k = 0; k < K; ++k) ch[k] = chars[start + idx[k]];
            int win = -1;
#pragma unroll
            for (int k = K - 1; k >= 0; --k)
                if (static_cast<uint32_t>(r[k]) <= s_tab.accept_le[s_tab.row_of[ch[k]]]) win = k;  // the FIRST accepted attempt
            if (win >= 0) {
                int64_t iw = idx[0];
                uint8_t cw = ch[0];
#pragma unroll
                for (int k = 1; k < K; ++k)
                    if (win == k) iw = idx[k], cw = ch[k];
                const uint32_t c = ctr + static_cast<uint32_t>(win);
                const int row = s_tab.row_of[cw];
                const double pself = s_tab.self[row];
                const double *cdf = s_tab.cdf[row];
                const double u = unit(mix64(h0 + 0xD1342543DE82EF95ull * (static_cast<uint64_t>(c) + 2))) * (cdf[kCols - 1] - pself);
                uint32_t below = 0;
#pragma unroll
                for (int q = 0; q < kCols; ++q) below |= static_cast<uint32_t>(q != row && u < cdf[q] - (q > row ? pself : 0.0)) << q;
                const int last = row == kCols - 1 ? kCols - 2 : kCols - 1;
                chars[start + iw] = s_tab.letter[below ? __builtin_ctz(below) : last];
                ctr = c + 2;
                rem -= 1;
            } else {
                ctr += K;
                tries = K;
            }
        }
        if (__builtin_amdgcn_ballot_w64(rem > 0) == 0) return;  // wave-uniform: every mutation of the wave is made
        s_start[wave][lane] = start;
        s_len[wave][lane] = L;
        s_h0[wave][lane] = h0;
        s_ctr[wave][lane] = ctr;
        s_rem[wave][lane] = rem;
        s_tries[wave][lane] = tries;
    }
    __shared__ int32_t s_sel[4][kSeqPerWave];
    for (;;) {
        // state written in the previous step (LDS: in order within a wave); a sequence that is visited AGAIN after a
        // mutation (chain_len > 1) must also see the character that was stored: wait for the stores then
        if (chain_len > 1) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const bool pending = s_rem[wave][lane] > 0;
        const uint64_t todo = __builtin_amdgcn_ballot_w64(pending);
        if (todo == 0) break;
        // A = attempts per sequence this step = 64 / (pending sequences rounded up to a power of two): 32 pending -> 2
        // attempts each, 16 -> 4, ..., 1 -> 64.  The home lane of the r-th pending sequence publishes itself in s_sel[r].
        const int npend = __builtin_popcountll(todo);
        const int groups = npend <= 1 ? 1 : 1 << (32 - __builtin_clz(static_cast<unsigned>(npend - 1)));  // wave-uniform
        const int shiftA = __builtin_ctz(64 / groups), A = 1 << shiftA;
        if (pending) s_sel[wave][__builtin_popcountll(todo & ((uint64_t(1) << lane) - 1))] = lane;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int g = lane >> shiftA, a = lane & (A - 1);
        const bool have = g < npend;
        const int sidx = have ? s_sel[wave][g] : 0;
        const int64_t start = s_start[wave][sidx], L = s_len[wave][sidx];
        const uint64_t h0 = s_h0[wave][sidx];
        const uint32_t ctr0 = s_ctr[wave][sidx];
        const int32_t tries = s_tries[wave][sidx];
        // K attempts per lane and round (round 3): attempt j = k * A + a has counter ctr0 + j.  Every position is a function
        // of (key, counter, length) alone, so the K gathers of a lane go out TOGETHER -- one memory round trip evaluates A * K
        // attempts of a sequence instead of A.  K = 1 is the round-2 kernel; results do not depend on K (the first accepted
        // attempt in counter order wins either way).  The kernel is latency-bound (59 % of its wave cycles are waits,
        // profiles/r03/augment_groups_pmc.txt): K = 4 takes the typical wave from 3-4 dependent rounds to 2, 21.8 -> 17.3-18.2 us.
        const bool long_len = __builtin_amdgcn_ballot_w64((static_cast<uint64_t>(L) >> 32) != 0) != 0;  // wave-uniform, never in practice
        uint64_t r[K];
        int64_t idx[K];
        uint8_t ch[K];
        bool valid[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t j = static_cast<uint32_t>(k * A + a);
            r[k] = mix64(h0 + 0xD1342543DE82EF95ull * (static_cast<uint64_t>(ctr0 + j) + 1));
            idx[k] = static_cast<int64_t>(mulhi_64x32(r[k], static_cast<uint32_t>(L)));  // uniform in [0, L)
            valid[k] = have && tries + static_cast<int32_t>(j) < kMaxAttempts;         // (attempts beyond the cap of this mutation are not made)
        }
        if (long_len) {  // a sequence of 2^32 characters or more in the wave: the full 64 x 64 multiply
#pragma unroll
            for (int k = 0; k < K; ++k) idx[k] = static_cast<int64_t>(__umul64hi(r[k], static_cast<uint64_t>(L)));
        }
#pragma unroll
        for (int k = 0; k < K; ++k) ch[k] = valid[k] ? chars[start + idx[k]] : uint8_t(0);
        int win_k = -1, win_a = 0;  // group-uniform: the first accepted attempt in counter order
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const bool accepted = valid[k] && static_cast<uint32_t>(r[k]) <= s_tab.accept_le[s_tab.row_of[ch[k]]];  // = double(lo32) * 2^-32 < 1 - p_self
            const uint64_t acc = __builtin_amdgcn_ballot_w64(accepted);
            const uint64_t mine = (acc >> (g << shiftA)) & (A == 64 ? ~uint64_t(0) : ((uint64_t(1) << A) - 1));  // this group's attempts
            if (win_k < 0 && mine != 0) {
                win_k = k;
                win_a = __builtin_ctzll(mine);
            }
        }
        if (have) {
            if (win_k >= 0) {
                if (a == win_a) {  // this lane made the winning attempt: draw the new residue, write it
                    int64_t iw = idx[0];
                    uint8_t cw = ch[0];
#pragma unroll
                    for (int k = 1; k < K; ++k)
                        if (win_k == k) iw = idx[k], cw = ch[k];
                    const uint32_t c = ctr0 + static_cast<uint32_t>(win_k * A + a);
                    const int row = s_tab.row_of[cw];
                    const double pself = s_tab.self[row];
                    const double *cdf = s_tab.cdf[row];
                    const double u = unit(mix64(h0 + 0xD1342543DE82EF95ull * (static_cast<uint64_t>(c) + 2))) * (cdf[kCols - 1] - pself);
                    // first q != row with u < cdf[q] - (q > row ? pself : 0), else the last q != row: all twenty comparisons at once
                    // (the early-exit loop paid one LDS round trip per step)
                    uint32_t below = 0;
#pragma unroll
                    for (int q = 0; q < kCols; ++q) below |= static_cast<uint32_t>(q != row && u < cdf[q] - (q > row ? pself : 0.0)) << q;
                    const int last = row == kCols - 1 ? kCols - 2 : kCols - 1;
                    const int pick = below ? __builtin_ctz(below) : last;
                    chars[start + iw] = s_tab.letter[pick];
                    s_ctr[wave][sidx] = c + 2;
                    s_rem[wave][sidx] -= 1;
                    s_tries[wave][sidx] = 0;
                }
            } else if (a == 0) {  // A * K rejections: the next counters, or give this mutation up at the cap like the twin
                const int32_t made = tries + A * K < kMaxAttempts ? A * K : kMaxAttempts - tries;
                s_ctr[wave][sidx] = ctr0 + static_cast<uint32_t>(made);
                if (tries + made >= kMaxAttempts) {
                    s_rem[wave][sidx] -= 1;
                    s_tries[wave][sidx] = 0;
                } else {
                    s_tries[wave][sidx] = tries + made;
                }
            }
        }
    }
}

AugTable *g_dev_table[16] = {};
std::mutex g_table_mu;

bsq_status device_table(AugTable **out) {
    std::lock_guard<std::mutex> lock(g_table_mu);
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return bsq_internal::set_hip_error("hipGetDevice", e);
    if (dev < 0 || dev >= 16) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "device ordinal out of range");
    if (!g_dev_table[dev]) {
        AugTable h;
        double nr[kRows * kCols];
        make_normrows(nr);
        for (int r = 0; r < kRows; ++r) {
            double acc = 0.0;
            for (int c = 0; c < kCols; ++c) {
                acc += nr[r * kCols + c];
                h.cdf[r][c] = acc;
            }
            h.self[r] = r < kCols ? nr[r * kCols + r] : 0.0;
            // lo * 2^-32 < t  <=>  lo < t * 2^32 (exact: scaling by a power of two)  <=>  lo <= ceil(t * 2^32) - 1
            const double t32 = std::ldexp(1.0 - h.self[r], 32);
            const double c = std::ceil(t32);
            h.accept_le[r] = c >= 4294967296.0 ? 0xFFFFFFFFu : static_cast<uint32_t>(static_cast<uint64_t>(c) - 1);  // (t > 0 always)
        }
        h.pad_[0] = h.pad_[1] = h.pad_[2] = h.pad_[3] = 0;
        std::memset(h.row_of, kRows - 1, sizeof(h.row_of));
        for (int c = 0; c < kCols; ++c) {
            h.row_of[static_cast<unsigned char>(kLetters[c])] = static_cast<uint8_t>(c);
            h.letter[c] = static_cast<uint8_t>(kLetters[c]);
        }
        AugTable *d = nullptr;
        e = hipMalloc(reinterpret_cast<void **>(&d), sizeof(AugTable));
        if (e != hipSuccess) return bsq_internal::set_hip_error("hipMalloc(augment table)", e);
        e = hipMemcpy(d, &h, sizeof(AugTable), hipMemcpyHostToDevice);
        if (e != hipSuccess) return bsq_internal::set_hip_error("hipMemcpy(augment table)", e);
        g_dev_table[dev] = d;
    }
    *out = g_dev_table[dev];
    return BSQ_OK;
}

}  // namespace

extern "C" {

bsq_status bsq_blosum62_normrows(double *out21x20) {
    if (!out21x20) return BSQ_ERR_INVALID_ARG;
    make_normrows(out21x20);
    return BSQ_OK;
}

bsq_status bsq_augment_device(uint8_t *chars, const int64_t *offsets, int64_t B, int32_t chain_len, double frac,
                              uint64_t seed, void *hip_stream) {
    if (!offsets || B < 0 || chain_len < 0) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "bad augment arguments");
    if (B == 0 || chain_len == 0 || !(frac > 0.0)) return BSQ_OK;
    if (!chars) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "chars is null");
    AugTable *tab = nullptr;
    const bsq_status st = device_table(&tab);
    if (st != BSQ_OK) return st;
    const int64_t blocks = (B + 255) / 256;
    if (blocks >= (int64_t(1) << 31)) return bsq_internal::set_error(BSQ_ERR_INVALID_ARG, "batch too large");
#ifdef BSQ_LABS
    if (bsq_internal::tuning().augment_mode == 1) {  // one lane per sequence (round 1)
        hipLaunchKernelGGL(k_augment, dim3(unsigned(blocks)), dim3(256), 0, static_cast<hipStream_t>(hip_stream), chars,
                           offsets, B, chain_len, frac, seed, tab);
        const hipError_t e1 = hipGetLastError();
        if (e1 != hipSuccess) return bsq_internal::set_hip_error("k_augment", e1);
        return BSQ_OK;
    }
#endif
    // knob "augment_k": attempts per lane and round (0 automatic = 4; 1 = the round-2 form; 2) -- speed only
    const int ak = bsq_internal::tuning().augment_k;
    if (ak == 1)
        hipLaunchKernelGGL(k_augment_groups<1>, dim3(unsigned(blocks)), dim3(256), 0, static_cast<hipStream_t>(hip_stream), chars,
                           offsets, B, chain_len, frac, seed, tab);
    else if (ak == 2)
        hipLaunchKernelGGL(k_augment_groups<2>, dim3(unsigned(blocks)), dim3(256), 0, static_cast<hipStream_t>(hip_stream), chars,
                           offsets, B, chain_len, frac, seed, tab);
    else
        hipLaunchKernelGGL(k_augment_groups<4>, dim3(unsigned(blocks)), dim3(256), 0, static_cast<hipStream_t>(hip_stream), chars,
                           offsets, B, chain_len, frac, seed, tab);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return bsq_internal::set_hip_error("k_augment", e);
    return BSQ_OK;
}

}  // extern "C"
