/*
 * TEST INFRASTRUCTURE ONLY -- see bsq_oracle.h.  Plain C11 + OpenMP restatement of the
 * reference's batch tokenizer / one-hot encoder, written from the behaviour described in
 * SURVEY.md section 8a; every function cites the reference lines it follows.
 */
#include "bsq_oracle.h"

#include <ctype.h>
#include <stddef.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * Alphabets.  Group strings and alias strings are the reference's data
 * (src/alphabet.h:108-124, :189-194); the key table is CAMAP (src/alphabet.h:198-222).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const char *key;
    const char *groups; /* NULL => BYTES identity table */
    const char *alias;
} bsqo_alpha;

#define PROT_ALIAS "OU:KC"
#define DNA_ALIAS "U:T"
/* kept in the order std::map<std::string,...> iterates (lexicographic) */
static const bsqo_alpha k_alpha[] = {
    {"AMINO", "A,C,D,E,F,G,H,I,K,L,M,N,P,Q,R,S,T,V,W,Y", PROT_ALIAS},
    {"AMINO20", "A,C,D,E,F,G,H,I,K,L,M,N,P,Q,R,S,T,V,W,Y", PROT_ALIAS},
    {"BYTES", NULL, NULL},
    {"C", "C,AGT", DNA_ALIAS},
    {"DAYHOFF", "AGPST,C,DENQ,FWY,HKR,ILMV", PROT_ALIAS},
    {"DNA", "A,C,G,T", DNA_ALIAS},
    {"DNA4", "A,C,G,T", DNA_ALIAS},
    {"DNA5", "A,C,G,T,NMRWSYKVHDB", DNA_ALIAS},
    {"DNAMETH", "C,AGT", DNA_ALIAS},
    {"KETO", "ACM,KGT", DNA_ALIAS},
    {"LIA10", "AC,DE,FWY,G,HN,IV,KQR,LM,P,ST", PROT_ALIAS},
    {"LIB10", "AST,C,DEQ,FWY,G,HN,IV,KR,LM,P", PROT_ALIAS},
    {"MURPHY", "A,C,DENQ,FWY,G,H,ILMV,KR,P,ST", PROT_ALIAS},
    {"PROTEIN", "A,C,D,E,F,G,H,I,K,L,M,N,P,Q,R,S,T,V,W,Y", PROT_ALIAS},
    {"PURPYR", "AGR,YCT", DNA_ALIAS},
    {"SEB10", "AST,C,DN,EQ,FY,G,HW,ILMV,KR,P", PROT_ALIAS},
    {"SEB14", "A,C,D,EQ,FY,G,H,IV,KR,LM,N,P,ST,W", PROT_ALIAS},
    {"SEB6", "AST,CP,DHNEKQR,FWY,G,ILMV", PROT_ALIAS},
    {"SEB8", "AST,C,DHN,EKQR,FWY,G,ILMV,P", PROT_ALIAS},
    {"SEV10", "AST,C,DEN,FY,G,H,ILMV,KQR,P,W", PROT_ALIAS},
};
enum { k_nalpha = (int)(sizeof(k_alpha) / sizeof(k_alpha[0])) };

int bsqo_num_keys(void) { return k_nalpha; }
const char *bsqo_key(int i) { return (i >= 0 && i < k_nalpha) ? k_alpha[i].key : NULL; }

/* src/alphabet.h:32-61.  Characters between commas share one id; both cases are set
 * (:39, :44).  The alias pass (:46-58) is restated literally, including its defect: the value
 * stored for an alias source is lut[ lut[target_char] ] -- the target's *id* is used as a byte
 * index, which lands on a control character (-1) for every shipped alphabet, so O/U stay
 * unmapped (probed against the compiled reference). */
static int build_lut(const char *groups, const char *alias, int8_t lut[256]) {
    memset(lut, 0xff, 256);
    int id = 0;
    for (const char *p = groups; *p; ++p) {
        if (*p == ',') {
            ++id;
            continue;
        }
        const unsigned char v = (unsigned char)*p;
        lut[v | 32u] = (int8_t)id;
        lut[v & 0xdfu] = (int8_t)id;
    }
    if (alias) {
        const char *colon = strchr(alias, ':');
        if (colon) {
            const size_t nsrc = (size_t)(colon - alias);
            for (size_t i = 0; i < nsrc; ++i) {
                const int8_t target_id = lut[(unsigned char)colon[i + 1]];
                /* target_id >= 0 for every shipped alias string (K, C, T are always mapped) */
                const int8_t stored = lut[(unsigned char)target_id];
                const unsigned char up = (unsigned char)alias[i] & 0xdfu, lo = (unsigned char)alias[i] | 32u;
                if (lut[up] == -1) lut[up] = stored;
                if (lut[lo] == -1) lut[lo] = stored;
            }
        }
    }
    return id + 1; /* nchars() = commas + 1 (src/alphabet.h:27) */
}

int bsqo_alphabet(const char *key, int8_t lut[256], int *nchars) {
    char up[32];
    size_t n = strlen(key);
    if (n >= sizeof(up)) return -1;
    for (size_t i = 0; i <= n; ++i) up[i] = (char)toupper((unsigned char)key[i]); /* tokenize.h:73 */
    for (int a = 0; a < k_nalpha; ++a) {
        if (strcmp(up, k_alpha[a].key) != 0) continue;
        if (k_alpha[a].groups == NULL) { /* BYTES: src/alphabet.h:91-97, lut[i] = int8(i) */
            for (int i = 0; i < 256; ++i) lut[i] = (int8_t)i;
            *nchars = 256;
        } else {
            *nchars = build_lut(k_alpha[a].groups, k_alpha[a].alias, lut);
        }
        return 0;
    }
    return -1;
}

/* src/tokenize.h:22-38 */
int bsqo_bos_id(int nchars, int eos, int bos, int padchar) { (void)eos; (void)padchar; return bos ? nchars : -1; }
int bsqo_eos_id(int nchars, int eos, int bos, int padchar) { (void)padchar; return eos ? nchars + (bos != 0) : -1; }
int bsqo_pad_id(int nchars, int eos, int bos, int padchar) { (void)padchar; return nchars + (bos != 0) + (eos != 0); }
int bsqo_alphabet_size(int nchars, int eos, int bos, int padchar) {
    return nchars + (eos != 0) + (bos != 0) + (padchar != 0);
}

/* src/tokenize.cpp:65-98: switch(std::tolower(dt[0])) -- the upper-case labels are dead. */
int bsqo_dtype_from_destchar(char c) {
    switch (tolower((unsigned char)c)) {
    case 'b': return BSQO_I8;
    case 'h': return BSQO_I16;
    case 'i': return BSQO_I32;
    case 'l':
    case 'q': return BSQO_U64;
    case 'f': return BSQO_F32;
    case 'd': return BSQO_F64;
    default: return -1;
    }
}
int bsqo_dtype_size(int dtype) {
    static const int sz[6] = {1, 2, 4, 8, 4, 8};
    return (dtype >= 0 && dtype < 6) ? sz[dtype] : 0;
}

/* Bytes >= 0x80: the reference indexes the table with a signed char (src/alphabet.h:78), which
 * reads before the array (UB).  The build defines them as unmapped; the oracle does the same and
 * must never be compared with the compiled reference on such input. */
static inline int translate(const int8_t lut[256], uint8_t c) { return c < 0x80 ? lut[c] : -1; }

static int64_t first_too_long(const int64_t *offsets, int64_t B, int64_t P, int extra) {
    for (int64_t i = 0; i < B; ++i)
        if (offsets[i + 1] - offsets[i] + extra > P) return i;
    return -1;
}

#define STORE_CASES(EXPR_IDX, VALUE)                                                               \
    switch (dtype) {                                                                               \
    case BSQO_I8: ((int8_t *)out)[EXPR_IDX] = (int8_t)(VALUE); break;                              \
    case BSQO_I16: ((int16_t *)out)[EXPR_IDX] = (int16_t)(VALUE); break;                           \
    case BSQO_I32: ((int32_t *)out)[EXPR_IDX] = (int32_t)(VALUE); break;                           \
    case BSQO_U64: ((uint64_t *)out)[EXPR_IDX] = (uint64_t)(VALUE); break;                         \
    case BSQO_F32: ((float *)out)[EXPR_IDX] = (float)(VALUE); break;                               \
    default: ((double *)out)[EXPR_IDX] = (double)(VALUE); break;                                   \
    }

/* src/tokenize.h:381-485.  memset(0) then, per sequence (OpenMP over sequences, :451-454):
 * BOS at 0, lut[c] at bos+j only when >= 0 (:438-447 `if(charind >= 0)`), EOS at bos+L,
 * PAD id from L+bos+eos to P when padchar.  Index is b*P+t (batch_first) or t*B+b. */
int64_t bsqo_tokenize(const int8_t lut[256], int nchars, int eos, int bos, int padchar,
                      const uint8_t *chars, const int64_t *offsets, int64_t B, int64_t P,
                      int batch_first, int dtype, void *out, int nthreads) {
    const int hb = bos != 0, he = eos != 0;
    const int64_t bad = first_too_long(offsets, B, P, hb + he);
    if (bad >= 0) return bad + 1;
    if (nthreads <= 0) nthreads = 1; /* :384 */
    const int bos_id = bsqo_bos_id(nchars, eos, bos, padchar);
    const int eos_id = bsqo_eos_id(nchars, eos, bos, padchar);
    const int pad_id = bsqo_pad_id(nchars, eos, bos, padchar);
    memset(out, 0, (size_t)B * (size_t)P * (size_t)bsqo_dtype_size(dtype)); /* :427 */
#define TOK_AT(t, b) (batch_first ? (size_t)(b) * (size_t)P + (size_t)(t) : (size_t)(t) * (size_t)B + (size_t)(b))
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int64_t i = 0; i < B; ++i) {
        const uint8_t *s = chars + offsets[i];
        const int64_t L = offsets[i + 1] - offsets[i];
        if (hb) { STORE_CASES(TOK_AT(0, i), bos_id) }
        for (int64_t j = 0; j < L; ++j) {
            const int tr = translate(lut, s[j]);
            if (tr >= 0) { STORE_CASES(TOK_AT(hb + j, i), tr) }
        }
        if (he) { STORE_CASES(TOK_AT(hb + L, i), eos_id) }
        if (padchar)
            for (int64_t k = L + hb + he; k < P; ++k) { STORE_CASES(TOK_AT(k, i), pad_id) }
    }
#undef TOK_AT
    return 0;
}

/* src/tokenize.h:283-371.  Single memset of the whole (P,B,C) tensor (:332), then per sequence
 * (OpenMP, :339-342): [0,i,bos]=1; [bos+j,i,lut[c]]=1 when (no mask or mask[j]) and lut[c]>=0
 * (:345-353); [bos+L,i,eos]=1; [k,i,pad]=1 for k>=L+bos+eos when padchar (:363-368).
 * Flat index (t*B + i)*C + ch (:327-330). */
int64_t bsqo_onehot(const int8_t lut[256], int nchars, int eos, int bos, int padchar,
                    const uint8_t *chars, const int64_t *offsets, const uint8_t *mask,
                    int64_t B, int64_t P, int dtype, void *out, int nthreads) {
    const int hb = bos != 0, he = eos != 0;
    const int64_t bad = first_too_long(offsets, B, P, hb + he);
    if (bad >= 0) return bad + 1;
    if (nthreads <= 0) nthreads = 1;
    const int bos_id = bsqo_bos_id(nchars, eos, bos, padchar);
    const int eos_id = bsqo_eos_id(nchars, eos, bos, padchar);
    const int pad_id = bsqo_pad_id(nchars, eos, bos, padchar);
    const size_t C = (size_t)bsqo_alphabet_size(nchars, eos, bos, padchar);
    memset(out, 0, (size_t)P * (size_t)B * C * (size_t)bsqo_dtype_size(dtype));
#define HOT(t, b, c) (((size_t)(t) * (size_t)B + (size_t)(b)) * C + (size_t)(c))
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int64_t i = 0; i < B; ++i) {
        const uint8_t *s = chars + offsets[i];
        const uint8_t *m = mask ? mask + offsets[i] : NULL;
        const int64_t L = offsets[i + 1] - offsets[i];
        if (hb) { STORE_CASES(HOT(0, i, bos_id), 1) }
        for (int64_t j = 0; j < L; ++j) {
            if (m && !m[j]) continue;
            const int tr = translate(lut, s[j]);
            if (tr >= 0) { STORE_CASES(HOT(hb + j, i, tr), 1) }
        }
        if (he) { STORE_CASES(HOT(hb + L, i, eos_id), 1) }
        if (padchar)
            for (int64_t k = L + hb + he; k < P; ++k) { STORE_CASES(HOT(k, i, pad_id), 1) }
    }
#undef HOT
    return 0;
}
