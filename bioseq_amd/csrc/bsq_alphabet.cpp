// Alphabet tables, tokenizer ids and dtype parsing of libbsq_hip.so (host side, no HIP calls).
//
// What this replaces in the reference: alph::TAlphabet::make_lut + the alphabet constants and
// CAMAP (/root/reference/src/alphabet.h:32-61, :108-124, :189-194, :198-222), the id accessors
// of struct Tokenizer (src/tokenize.h:21-38) and the dtype switch of src/tokenize.cpp:65-98.
//
// Design note: the reference declares alias strings ("OU:KC", "U:T") but its alias pass stores
// lut[lut[target]] (alphabet.h:52-56), i.e. the lookup of a control character, so the aliases
// are inert in the compiled tables (O/U stay unmapped).  The tables here
// are therefore built from the comma groups alone; tests/test_oracle_golden.py and tests/test_host_surface.py pin
// all 20 keys against tests/golden/alphabets.json (LUTs dumped from the compiled reference).
#include "bsq.h"
#include "bsq_diag.h"

#include <cctype>
#include <cstring>
#include <string>

namespace {

struct Spec {
    const char *key;
    const char *groups;  // nullptr: identity byte table (BYTES)
};

constexpr const char *kAmino20 = "A,C,D,E,F,G,H,I,K,L,M,N,P,Q,R,S,T,V,W,Y";
constexpr const char *kDna4 = "A,C,G,T";
constexpr const char *kMeth = "C,AGT";

// Sorted by key: the order std::map<std::string, ...> (CAMAP) lists them in its error message.
constexpr Spec kSpecs[] = {
    {"AMINO", kAmino20},
    {"AMINO20", kAmino20},
    {"BYTES", nullptr},
    {"C", kMeth},
    {"DAYHOFF", "AGPST,C,DENQ,FWY,HKR,ILMV"},
    {"DNA", kDna4},
    {"DNA4", kDna4},
    {"DNA5", "A,C,G,T,NMRWSYKVHDB"},
    {"DNAMETH", kMeth},
    {"KETO", "ACM,KGT"},
    {"LIA10", "AC,DE,FWY,G,HN,IV,KQR,LM,P,ST"},
    {"LIB10", "AST,C,DEQ,FWY,G,HN,IV,KR,LM,P"},
    {"MURPHY", "A,C,DENQ,FWY,G,H,ILMV,KR,P,ST"},
    {"PROTEIN", kAmino20},
    {"PURPYR", "AGR,YCT"},
    {"SEB10", "AST,C,DN,EQ,FY,G,HW,ILMV,KR,P"},
    {"SEB14", "A,C,D,EQ,FY,G,H,IV,KR,LM,N,P,ST,W"},
    {"SEB6", "AST,CP,DHNEKQR,FWY,G,ILMV"},
    {"SEB8", "AST,C,DHN,EKQR,FWY,G,ILMV,P"},
    {"SEV10", "AST,C,DEN,FY,G,H,ILMV,KQR,P,W"},
};
constexpr int kNumSpecs = int(sizeof(kSpecs) / sizeof(kSpecs[0]));

const Spec *find_spec(const char *key) {
    if (!key) return nullptr;
    std::string up(key);
    for (char &c : up) c = char(std::toupper(static_cast<unsigned char>(c)));
    for (const Spec &s : kSpecs)
        if (up == s.key) return &s;
    return nullptr;
}

int fill_table(const Spec &s, int8_t lut[256]) {
    if (!s.groups) {  // BYTES (alphabet.h:91-97): lut[i] = int8(i); ids >= 128 read back negative = unmapped
        for (int i = 0; i < 256; ++i) lut[i] = static_cast<int8_t>(static_cast<uint8_t>(i));
        return 256;
    }
    std::memset(lut, -1, 256);
    int group = 0;
    for (const char *p = s.groups; *p; ++p) {
        if (*p == ',') {
            ++group;
        } else {  // letters only: set both cases
            lut[static_cast<unsigned char>(std::toupper(*p))] = int8_t(group);
            lut[static_cast<unsigned char>(std::tolower(*p))] = int8_t(group);
        }
    }
    return group + 1;
}

}  // namespace

extern "C" {

int32_t bsq_abi_version(void) { return BSQ_ABI_VERSION; }
#ifndef BSQ_BUILD_ID
#define BSQ_BUILD_ID "unstamped"
#endif
const char *bsq_build_id(void) { return BSQ_BUILD_ID; }

const char *bsq_strerror(bsq_status s) {
    switch (s) {
    case BSQ_OK: return "ok";
    case BSQ_ERR_INVALID_KEY: return "Invalid tokenizer type";
    case BSQ_ERR_INVALID_ARG: return "invalid argument";
    case BSQ_ERR_DTYPE: return "Unsupported dtype";
    case BSQ_ERR_SEQ_TOO_LONG: return "seq len + bos + eos > padlen";
    case BSQ_ERR_NO_DEVICE: return "no HIP device available (bioseq_amd has no CPU fallback)";
    case BSQ_ERR_HIP: return "HIP runtime error";
    case BSQ_ERR_ALLOC: return "allocation failed";
    case BSQ_ERR_FUSED_WAIT: return "a fused augmentation + token launch gave up waiting inside the kernel (its output is poisoned with 0xFF)";
    default: return "unknown bsq_status";
    }
}

int32_t bsq_num_keys(void) { return kNumSpecs; }
const char *bsq_key_name(int32_t i) { return (i >= 0 && i < kNumSpecs) ? kSpecs[i].key : nullptr; }

bsq_status bsq_lut_get(const char *key, int8_t lut[256], int32_t *nchars) {
    if (!lut || !nchars) return BSQ_ERR_INVALID_ARG;
    const Spec *s = find_spec(key);
    if (!s) return BSQ_ERR_INVALID_KEY;
    *nchars = fill_table(*s, lut);
    return BSQ_OK;
}

bsq_status bsq_desc_init(bsq_desc *d, const char *key, int32_t eos, int32_t bos, int32_t padchar) {
    if (!d) return BSQ_ERR_INVALID_ARG;
    const bsq_status st = bsq_lut_get(key, d->lut, &d->nchars);
    if (st != BSQ_OK) return st;
    d->eos = eos != 0;
    d->bos = bos != 0;
    d->padchar = padchar != 0;
    return BSQ_OK;
}

int32_t bsq_bos_id(const bsq_desc *d) { return d->bos ? d->nchars : -1; }
int32_t bsq_eos_id(const bsq_desc *d) { return d->eos ? d->nchars + d->bos : -1; }
int32_t bsq_pad_id(const bsq_desc *d) { return d->nchars + d->bos + d->eos; }
int32_t bsq_alphabet_size(const bsq_desc *d) { return d->nchars + d->eos + d->bos + d->padchar; }

bsq_status bsq_dtype_from_destchar(char c, bsq_dtype *out) {
    if (!out) return BSQ_ERR_INVALID_ARG;
    switch (std::tolower(static_cast<unsigned char>(c))) {
    case 'b': *out = BSQ_I8; return BSQ_OK;
    case 'h': *out = BSQ_I16; return BSQ_OK;
    case 'i': *out = BSQ_I32; return BSQ_OK;
    case 'l':
    case 'q': *out = BSQ_U64; return BSQ_OK;
    case 'f': *out = BSQ_F32; return BSQ_OK;
    case 'd': *out = BSQ_F64; return BSQ_OK;
    default: return BSQ_ERR_DTYPE;
    }
}

size_t bsq_dtype_size(bsq_dtype t) {
    switch (t) {
    case BSQ_I8: return 1;
    case BSQ_I16: return 2;
    case BSQ_I32:
    case BSQ_F32: return 4;
    case BSQ_U64:
    case BSQ_F64: return 8;
    }
    return 0;
}

bsq_status bsq_validate_lengths(const int64_t *offsets, int64_t B, int64_t P, int32_t bos, int32_t eos,
                                int64_t *first_bad) {
    if (!offsets || B < 0 || !first_bad) return BSQ_ERR_INVALID_ARG;
    *first_bad = -1;
    if (P <= 0) return BSQ_ERR_INVALID_ARG;
    const int64_t room = P - (bos != 0) - (eos != 0);
    for (int64_t i = 0; i < B; ++i) {
        if (offsets[i + 1] - offsets[i] > room) {
            *first_bad = i;
            return BSQ_ERR_SEQ_TOO_LONG;
        }
    }
    return BSQ_OK;
}

}  // extern "C"
