// Development and test switches of the library, in ONE place (r06).
//
// The product library (dicey_amd/libdiceygpu.so, `make`) has one behaviour: exp_env() is a constant nullptr there and every branch
// behind it is compiled out.  What it still reads from the environment is deployment, not behaviour: DICEY_TIMING (phase times on
// stderr), DICEY_HOST_THREADS, DICEY_CAP_BUDGET_MB, DICEY_KMER_K / DICEY_KMER_K2 (table and filter orders: tuning knobs, also how the
// tests give a 30 kb genome a table at all) and DICEY_FM9_HASH (the builder's stamp).
//
// A development build (`tools/build_variant.sh exp -DDG_EXPERIMENTS` -> dicey_amd/variants/libdiceygpu_exp.so; tests/conftest.py builds
// it when missing) reads the switches below.  They FORCE code paths the product takes by itself when the data asks for them — the
// search without the select stage inside (strings above 42 characters), the full-matrix verify (queries above 32 nt), the host
// enumerator of capped neighbourhoods (sequences with N), tiny buffer capacities (the retry path), an index without one of the
// derived layouts (a device short of memory) ... — so that the GPU suite can hold every path against the checker on the same
// small inputs (tests/test_gpu_parity.py SWITCHES), and a few measurement aids whose results are deliberately wrong (DICEY_EXP).
//   hunt.hip      DICEY_NO_BAND_VERIFY DICEY_CAP_HOST DICEY_NO_FUSED_SELECT DICEY_NO_FUSED_SELECT2 DICEY_NO_PREP_FUSION DICEY_NO_PRE5_D2
//                 DICEY_NO_FLAT_HAMMING2 DICEY_NO_N_WINDOW DICEY_NO_LONG2 DICEY_NO_DIRECT_CTX DICEY_DEBUG_CAPS DICEY_FUSED_LCAP DICEY_VERIFY_CH DICEY_DUMP_JOBS DICEY_EXP
//   index.hip     DICEY_NO_KMER_FILTER DICEY_NO_NRUN_PRUNE DICEY_NO_PRE5 DICEY_NO_SAX DICEY_NO_PLV DICEY_NO_SA_MINIMA DICEY_EXP_PRIO
//   search.hip / thal_api.hip   DICEY_NO_LDS_TABLES DICEY_NO_WAVE_THAL DICEY_DEBUG_THAL_REDO DICEY_DEBUG_DUMP_RAW
#pragma once
#include <cstdlib>

namespace dg {
#ifdef DG_EXPERIMENTS
inline const char* exp_env(const char* name) { return std::getenv(name); }
#else
inline const char* exp_env(const char*) { return nullptr; }
#endif
}  // namespace dg
