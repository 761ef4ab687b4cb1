// one translation unit per evaluation variant of glibc's sinf / cosf (urh_amd/csrc/glibc_sincosf.h): -DURH_SINCOSF_FMA=0 / 1, -DNAME=...
#include "../../urh_amd/csrc/glibc_sincosf.h"
#define CAT(a, b) a##b
#define X(a, b) CAT(a, b)
float X(NAME, _sin)(float x) { return urh_sinf(x); }
float X(NAME, _cos)(float x) { return urh_cosf(x); }
