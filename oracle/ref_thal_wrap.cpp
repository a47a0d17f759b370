// ORACLE — TEST INFRASTRUCTURE ONLY.
// Thin C entry points around the UNMODIFIED reference header /root/reference/src/thal.h (primer3's thal.c as dicey
// ships it).  The header is compiled where it lies (-I/root/reference/src); nothing of it is copied into this repo.
// Output goes to oracle/_ref/libthalref.so, which is git-ignored and travels to the GPU box as a built artefact.
// Call sites this stands in for: silica.h:316-329 (init), silica.h:437 and silica.h:511 (thal per primer / per hit).
#include <cerrno>
#include <iostream>
#include <string>

#include <thal.h>

extern "C" {

static primer3thal::thal_args g_args;

// silica.h:316-329: defaults, temponly, thal_end1, tables from `config_dir` (must end with '/'), temp in Celsius
int ref_thal_init(const char* config_dir, double temp_c, double mv, double dv, double dna_conc, double dntp) {
  primer3thal::set_thal_default_args(&g_args);
  g_args.temponly = 1;
  g_args.type = primer3thal::thal_end1;
  if (primer3thal::get_thermodynamic_values(config_dir) != 0) return -1;
  g_args.temp = temp_c;
  g_args.mv = mv;
  g_args.dv = dv;
  g_args.dna_conc = dna_conc;
  g_args.dntp = dntp;
  g_args.temp += primer3thal::ABSOLUTE_ZERO;
  return 0;
}

// returns 1 on success (thal() returned true); temp = o.temp (THAL_ERROR_SCORE = -999999 on failure)
int ref_thal(const char* oligo1, const char* oligo2, double* temp, int* end1, int* end2) {
  primer3thal::thal_results o;
  bool ok = primer3thal::thal((const unsigned char*)oligo1, (const unsigned char*)oligo2, &g_args, &o);
  *temp = o.temp;
  *end1 = o.align_end_1;
  *end2 = o.align_end_2;
  return ok ? 1 : 0;
}

}  // extern "C"
