// TEST INFRASTRUCTURE ONLY -- not product code, never imported by bioseq_amd/.
//
// Module-init driver used to compile the reference's own tokenizer sources
// (/root/reference/src/tokenize.cpp + omp.cpp, read in place, never copied)
// into a throw-away `cbioseq` oracle under oracle/_ref/.
//
// The reference's own module init (/root/reference/src/bioseq.cpp:6-11) also
// registers fxstats (needs zlib + kseq) and poa (needs the spoa submodule,
// which is EMPTY in the reference checkout), so the whole extension cannot be
// built here.  This file registers the init functions of the hot path and of FlatFile (the
// path's packed-batch source); all are defined by the reference sources themselves.
#include "bioseq.h"   // from -I/root/reference/src
void init_omp_helpers(py::module &m);
void init_fxstats(py::module &m);  // FlatFile (src/fxstats.cpp; needs zlib + the vendored kseq/mio/span headers)
PYBIND11_MODULE(cbioseq, m) {
    init_tokenize(m);
    init_omp_helpers(m);
    init_fxstats(m);
}
