"""urh_amd/csrc/glibc_sincosf.h (the restatement the Costas and modulation kernels use) is bit-identical to the host
libm's sinf / cosf -- the functions the reference calls (signal_functions.pyx:165-166, :301) -- on small, medium and
large (reduce_large) arguments.  The FMA build of glibc (selected by ifunc on every host with FMA + AVX2) contracts
a + b*c: when the host has no FMA the header's non-FMA evaluation is compared instead."""
import os
import subprocess
import tempfile

from conftest import ROOT

SRC = r'''
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <stdint.h>
#define URH_SINCOSF_FMA %d
#include "%s"
static inline uint64_t sm(uint64_t *s){ uint64_t z=(*s+=0x9e3779b97f4a7c15ULL); z=(z^(z>>30))*0xbf58476d1ce4e5b9ULL; z=(z^(z>>27))*0x94d049bb133111ebULL; return z^(z>>31);}
static inline float u2f(uint32_t u){ float f; memcpy(&f,&u,4); return f; }
int main(int argc,char**argv){ long n=atol(argv[1]); long bad=0;
  #pragma omp parallel for reduction(+:bad)
  for(long t=0;t<64;t++){ uint64_t s=1234+t*7919;
    for(long i=0;i<n/64;i++){ uint64_t r=sm(&s); uint32_t a=(uint32_t)r, b=(uint32_t)(r>>32); float x; int mode=i&3;
      if(mode==0) x=u2f(a);                                              /* any bit pattern */
      else if(mode==1) x=((int32_t)a)/2147483648.0f*130.0f;              /* around the fast / large switch */
      else if(mode==2) x=((int32_t)a)/2147483648.0f*2.0e5f;              /* the carrier arguments of modulate_c */
      else x=u2f((a&0x807fffffu)|(((b>>7)%%60+120)<<23));               /* 2^-7 .. 2^52 */
      float r1=sinf(x), r2=urh_sinf(x); if(memcmp(&r1,&r2,4)!=0 && !(r1!=r1 && r2!=r2)) bad++;
      float q1=cosf(x), q2=urh_cosf(x); if(memcmp(&q1,&q2,4)!=0 && !(q1!=q1 && q2!=q2)) bad++; }}
  /* the branch-free pair of the Costas loop against the branchy functions: every 5th float below 120, every float below 2^-11,
     around 0.75 (the first branch's limit) and around the multiples of pi/4 */
  long bad2=0;
  #pragma omp parallel for reduction(+:bad2) schedule(dynamic,1)
  for(long blk=0;blk<2*(0x42f00000L>>16);blk++){ uint32_t sign=(blk&1)?0x80000000u:0u; uint32_t base=(uint32_t)(blk>>1)<<16;
    for(uint32_t i=0;i<65536;i++){ uint32_t u=base+i; float y=u2f(u|sign); float ay=fabsf(y);
      int crit = ay<0x1p-11f || (ay>0.7499f && ay<0.7501f);
      for(int q=1;q<=8 && !crit;q++){ float m=(float)(q*0.78539816339744830962); if(fabsf(ay-m)<2e-4f*m) crit=1; }
      if(!crit && (u%%5)!=0) continue;
      float sn,cs; urh_sincosf_fast(y,&sn,&cs); float r1=urh_sinf(y), r2=urh_cosf(y);
      if(memcmp(&sn,&r1,4)!=0 || memcmp(&cs,&r2,4)!=0) bad2++; }}
  printf("%%ld %%ld\n",bad,bad2); return 0; }
'''


def host_has_fma():
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        return True
    return " fma " in flags and " avx2 " in flags


def test_port_equals_libm():
    hdr = os.path.join(ROOT, "urh_amd", "csrc", "glibc_sincosf.h")
    fma = 1 if host_has_fma() else 0
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "chk.c")
        open(c, "w").write(SRC % (fma, hdr))
        exe = os.path.join(d, "chk")
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fopenmp"] + (["-mfma"] if fma else []) + [c, "-o", exe, "-lm"])
        out = subprocess.check_output([exe, "64000000"]).decode().strip()
    assert out.split() == ["0", "0"], f"{out}: mismatches against libm / of the branch-free pair against the branchy functions"


def test_host_libm_guard_reports_this_host():
    """urhgpu_host_libm_check (host arithmetic, no GPU): the device code restates glibc's FMA build of sinf / cosf and fdlibm's atan2f; on a
    host with FMA (every x86-64 box this runs on) the reference calls exactly those -- zero mismatches over the probes, which include the
    arguments on which glibc's two builds differ; a host without FMA is the case the guard exists for and must report mismatches."""
    from urh_amd import _lib
    v = _lib.host_libm_check()
    assert v["sincosf_compared"] >= 8000 and v["atan2f_compared"] >= 4000, v
    has_fma = False
    try:
        with open("/proc/cpuinfo") as fh:
            has_fma = any(line.startswith("flags") and " fma " in line + " " for line in fh)
    except OSError:
        pass
    assert v["atan2f_mismatches"] == 0, v
    if has_fma:
        assert v["sincosf_mismatches"] == 0, v
    else:
        assert v["sincosf_mismatches"] > 0, v
    assert _lib.host_libm_verdict(warn=False) == v
