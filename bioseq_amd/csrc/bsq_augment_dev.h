// Device side of the BLOSUM62 augmentation (bsq_augment.hip), shared with the fused augmentation + token launch of
// bsq_tokens8.hip: the table layout, the counter RNG and the body of k_augment_groups as a device function of a VIRTUAL block index.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace bsq_aug {

constexpr int kRows = 21, kCols = 20;

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
__device__ __forceinline__ double unit(uint64_t x) { return static_cast<double>(x >> 11) * 0x1.0p-53; }
// floor(r * len / 2^64) for len < 2^32: two 32 x 32 multiplies instead of the four of __umul64hi (quarter-rate instructions)
__device__ __forceinline__ uint64_t mulhi_64x32(uint64_t r, uint32_t len) {
    const uint64_t lo = static_cast<uint64_t>(static_cast<uint32_t>(r)) * len;
    return (static_cast<uint64_t>(static_cast<uint32_t>(r >> 32)) * len + (lo >> 32)) >> 32;
}

constexpr int kMaxAttempts = 1 << 14;  // the reference's `while inchar == outchar` is unbounded (an all-W sequence accepts with p = 0.006 per try)

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
// (virtual block `vblock` = 256 consecutive sequences; all 256 threads of a workgroup call it together: it holds a __syncthreads)
// COHERENT (the fused launch of bsq_tokens8.hip): the mutated characters are stored at AGENT scope -- written through to where the
// other XCDs' loads find them -- because waves of the SAME launch read them next.
// MAPPED (round 5, the same-XCD form of the fused launch): the lane's sequence is `b_mapped` (>= B: none) instead of the vblock's
// 256 consecutive ones -- the stream is keyed by (seed, sequence), so WHICH lane mutates a sequence never changes a result.
// RECORD (round 5, the no-wait form of the fused launch): every mutation is also written down -- rec[b * chain_len + m] = (position in
// the sequence, new byte) for the m-th mutation of sequence b, (0xFFFFFFFF, 0) where there is none -- so that a patch pass can put the
// mutated residues' tokens into a token matrix that was encoded WHILE the mutations were being made (bsq_tokens8.hip: k_patch_tokens).
// (records are stored at agent scope = written through: a line left dirty in L2 costs its write-back at the END of the launch, and the
//  patch launch behind it reads them from the memory side at once; measured neutral against plain stores, profiles/r05/aug_nowait_patch_ab.txt)
__device__ __forceinline__ void store_record(uint2 *dst, uint32_t pos, uint32_t byte) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(dst), (static_cast<unsigned long long>(byte) << 32) | pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int K, bool COHERENT = false, bool MAPPED = false, bool RECORD = false>
__device__ __forceinline__ void augment_groups_body(uint32_t vblock, uint8_t *chars, const int64_t *offsets, int64_t B, int32_t chain_len,
                                                    double frac, uint64_t seed, const AugTable *tab, int64_t b_mapped = 0, uint2 *rec = nullptr) {
    __shared__ __align__(16) AugTable s_tab;
    __shared__ int64_t s_start[4][kSeqPerWave], s_len[4][kSeqPerWave];
    __shared__ uint64_t s_h0[4][kSeqPerWave];
    __shared__ uint32_t s_ctr[4][kSeqPerWave];
    __shared__ int32_t s_rem[4][kSeqPerWave], s_tries[4][kSeqPerWave];
    __shared__ int64_t s_b[RECORD ? 4 : 1][RECORD ? kSeqPerWave : 1];  // RECORD: the sequence of every home lane (the winning lane of a group writes its record)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t b = MAPPED ? b_mapped : (static_cast<int64_t>(vblock) * 4 + wave) * kSeqPerWave + lane;
    if constexpr (RECORD) {
        if (b < B)
            for (int32_t m = 0; m < chain_len; ++m) store_record(rec + b * chain_len + m, 0xFFFFFFFFu, 0u);
    }
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
                if constexpr (COHERENT) __hip_atomic_store(chars + start + iw, static_cast<uint8_t>(s_tab.letter[below ? __builtin_ctz(below) : last]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else chars[start + iw] = s_tab.letter[below ? __builtin_ctz(below) : last];
                if constexpr (RECORD)  // the sequence's first mutation (positions of 2^32 and beyond lie in no token row: left as "none")
                    if (iw < 0xFFFFFFFFll) store_record(rec + b * chain_len, static_cast<uint32_t>(iw), static_cast<uint32_t>(s_tab.letter[below ? __builtin_ctz(below) : last]));
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
        if constexpr (RECORD) s_b[wave][lane] = b;
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
                    if constexpr (COHERENT) __hip_atomic_store(chars + start + iw, static_cast<uint8_t>(s_tab.letter[pick]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else chars[start + iw] = s_tab.letter[pick];
                    if constexpr (RECORD) {  // mutation number chain_len - (mutations still to make) of the home lane's sequence
                        const int64_t bs = s_b[wave][sidx];
                        if (bs < B && iw < 0xFFFFFFFFll)
                            store_record(rec + bs * chain_len + (chain_len - s_rem[wave][sidx]), static_cast<uint32_t>(iw), static_cast<uint32_t>(s_tab.letter[pick]));
                    }
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

}  // namespace bsq_aug
