// EXPERIMENT RECORD (round 3) -- not built into libbsq_hip.so.  The LDS-staged form of the BLOSUM62 rejection sampler
// (drop-in next to k_augment_groups in bioseq_amd/csrc/bsq_augment.hip, same tables and helpers).  Bit-identical results,
// SLOWER: profiles/r03/augment_staged_experiment.txt.
// LDS-STAGED form (round 3) of the same algorithm, the same random stream and the same results as k_augment_groups (the numpy
// twin in tests/test_augment.py judges all three).  Counters of k_augment_groups on the cfg5 batch
// (profiles/r03/augment_groups_pmc.txt): 59 % of the wave cycles are waits -- every round of attempts is a dependent,
// scattered one-byte gather out of global memory (~1.2 us each, 3-4 rounds per wave) -- and only 15 % vector work.  Here a
// wave owns 16 sequences and first copies the SELECTED ones into LDS with coalesced 16-byte loads, all in flight together
// (SLOT bytes per sequence, chosen by the host from the caller's length hint: 256 / 512 / 1024); the attempts then read
// their characters at LDS latency, and with 16 instead of 64 sequences per wave the first round already makes 64 / 8 = 8
// attempts for each of the ~8 selected ones (97 % are through after it).  A sequence longer than SLOT is simply not
// staged: its attempts gather from global memory as before, so the hint is only ever a speed matter.
constexpr int kStagedSeqPerWave = 16;
typedef uint32_t aug_u32x4u __attribute__((ext_vector_type(4), aligned(1)));
template <int SLOT>
__global__ __launch_bounds__(256) void k_augment_staged(uint8_t *chars, const int64_t *offsets, int64_t B, int32_t chain_len,
                                                        double frac, uint64_t seed, const AugTable *tab) {
    constexpr int SPW = kStagedSeqPerWave;
    constexpr int LPS = SLOT / 16;   // lanes that copy one sequence, 16 bytes each
    constexpr int SPI = 64 / LPS;    // sequences per copy instruction
    constexpr int NI = SPW / SPI;    // copy instructions per wave
    __shared__ AugTable s_tab;
    __shared__ __align__(16) uint8_t s_chars[4][SPW][SLOT];
    __shared__ int64_t s_start[4][SPW], s_len[4][SPW];
    __shared__ uint64_t s_h0[4][SPW];
    __shared__ uint32_t s_ctr[4][SPW];
    __shared__ int32_t s_rem[4][SPW], s_tries[4][SPW];
    __shared__ int32_t s_sel[4][SPW];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t b = (static_cast<int64_t>(blockIdx.x) * 4 + wave) * SPW + lane;
    // home lanes (lane < SPW): the spans first -- their loads fly while the table is staged
    int64_t start = 0, L = 0;
    if (lane < SPW && b < B) {
        start = offsets[b];
        L = offsets[b + 1] - start;
    }
    const int64_t total_chars = offsets[B];
    for (int i = threadIdx.x; i < int(sizeof(AugTable) / 4); i += 256)
        reinterpret_cast<uint32_t *>(&s_tab)[i] = reinterpret_cast<const uint32_t *>(tab)[i];
    __syncthreads();
    if (lane < SPW) {  // which sequences are augmented at all (word 0 of their stream), their keys
        int32_t rem = 0;
        uint64_t h0 = 0;
        if (b < B) {
            h0 = mix64(seed + 0x9E3779B97F4A7C15ull * (static_cast<uint64_t>(b) + 1));
            const bool pick = L > 0 && (!(frac < 1.0) || unit(mix64(h0 + 0xD1342543DE82EF95ull)) < frac);
            rem = pick ? chain_len : 0;
        }
        s_start[wave][lane] = start;
        s_len[wave][lane] = L;
        s_h0[wave][lane] = h0;
        s_ctr[wave][lane] = 1;
        s_rem[wave][lane] = rem;
        s_tries[wave][lane] = 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {   // stage the selected sequences that fit a slot: NI x (64 lanes x 16 bytes), every load issued before the first write
        const int sub = lane % LPS;
        aug_u32x4u v[NI];
        bool on[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int sidx = j * SPI + lane / LPS;
            const int64_t st = s_start[wave][sidx], len = s_len[wave][sidx];
            on[j] = s_rem[wave][sidx] > 0 && len <= SLOT && sub * 16 < len;
            v[j] = aug_u32x4u{0, 0, 0, 0};
            if (on[j]) {
                const int64_t a = st + sub * 16;
                if (a + 16 <= total_chars) {
                    v[j] = *reinterpret_cast<const aug_u32x4u *>(chars + a);
                } else {  // the last bytes of the whole buffer: never read past its end
                    uint32_t w[4] = {0, 0, 0, 0};
                    for (int i = 0; a + i < total_chars; ++i) w[i >> 2] |= static_cast<uint32_t>(chars[a + i]) << (8 * (i & 3));
                    v[j] = aug_u32x4u{w[0], w[1], w[2], w[3]};
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NI; ++j)
            if (on[j]) *reinterpret_cast<uint4 *>(&s_chars[wave][j * SPI + lane / LPS][sub * 16]) = uint4{v[j].x, v[j].y, v[j].z, v[j].w};
    }
    for (;;) {
        // state and staged characters written in the previous step (LDS: in order within a wave); an UNSTAGED sequence that is
        // visited again after a mutation (chain_len > 1) must also see the character that was stored: wait for the stores then
        if (chain_len > 1) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const bool pending = lane < SPW && s_rem[wave][lane] > 0;
        const uint64_t todo = __builtin_amdgcn_ballot_w64(pending);
        if (todo == 0) break;
        // A = attempts per sequence this step = 64 / (pending sequences rounded up to a power of two): 8 pending -> 8
        // attempts each, ..., 1 -> 64.  The home lane of the r-th pending sequence publishes itself in s_sel[r].
        const int npend = __builtin_popcountll(todo);
        const int groups = npend <= 1 ? 1 : 1 << (32 - __builtin_clz(static_cast<unsigned>(npend - 1)));  // wave-uniform
        const int shiftA = __builtin_ctz(64 / groups), A = 1 << shiftA;
        if (pending) s_sel[wave][__builtin_popcountll(todo & ((uint64_t(1) << lane) - 1))] = lane;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int g = lane >> shiftA, a = lane & (A - 1);
        const bool have = g < npend;
        const int sidx = have ? s_sel[wave][g] : 0;
        const int64_t st = s_start[wave][sidx], len = s_len[wave][sidx];
        const uint64_t h0 = s_h0[wave][sidx];
        const uint32_t c = s_ctr[wave][sidx] + static_cast<uint32_t>(a);  // counter of this lane's attempt
        const int32_t tries = s_tries[wave][sidx];
        const uint64_t r = mix64(h0 + 0xD1342543DE82EF95ull * (static_cast<uint64_t>(c) + 1));
        const int64_t idx = static_cast<int64_t>(__umul64hi(r, static_cast<uint64_t>(len)));  // uniform in [0, len)
        const bool staged = len <= SLOT;
        bool accepted = false;
        int row = 0;
        double pself = 0.0;
        if (have && tries + a < kMaxAttempts) {  // (attempts beyond the cap of this mutation are not made)
            const uint8_t ch = staged ? s_chars[wave][sidx][idx] : chars[st + idx];
            row = s_tab.row_of[ch];
            pself = s_tab.self[row];
            accepted = static_cast<double>(static_cast<uint32_t>(r)) * 0x1.0p-32 < 1.0 - pself;
        }
        const uint64_t acc = __builtin_amdgcn_ballot_w64(accepted);
        const uint64_t mine = (acc >> (g << shiftA)) & (A == 64 ? ~uint64_t(0) : ((uint64_t(1) << A) - 1));  // this group's attempts
        if (have) {
            if (mine != 0) {
                if (a == __builtin_ctzll(mine)) {  // first accepted attempt in counter order: draw the new residue, write it
                    const double *cdf = s_tab.cdf[row];
                    const double u = unit(mix64(h0 + 0xD1342543DE82EF95ull * (static_cast<uint64_t>(c) + 2))) * (cdf[kCols - 1] - pself);
                    int pick = -1;
                    for (int k = 0; k < kCols; ++k) {
                        if (k == row) continue;
                        pick = k;
                        if (u < cdf[k] - (k > row ? pself : 0.0)) break;
                    }
                    const uint8_t newc = s_tab.letter[pick];
                    chars[st + idx] = newc;
                    if (staged) s_chars[wave][sidx][idx] = newc;
                    s_ctr[wave][sidx] = c + 2;
                    s_rem[wave][sidx] -= 1;
                    s_tries[wave][sidx] = 0;
                }
            } else if (a == 0) {  // A rejections: the next counters, or give this mutation up at the cap like the twin
                const int32_t made = tries + A < kMaxAttempts ? A : kMaxAttempts - tries;
                s_ctr[wave][sidx] = c + static_cast<uint32_t>(made);
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

