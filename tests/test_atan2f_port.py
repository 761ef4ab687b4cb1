"""urh_amd/csrc/fdlibm_atan2f.h (the restatement the HIP kernel uses) is bit-identical to the host
libm's atan2f/atanf -- the functions the reference calls (signal_functions.pyx:376, C++ overload)."""
import os
import subprocess
import tempfile

from conftest import ROOT

SRC = r'''
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include "%s"
static inline uint64_t sm(uint64_t *s){ uint64_t z=(*s+=0x9e3779b97f4a7c15ULL); z=(z^(z>>30))*0xbf58476d1ce4e5b9ULL; z=(z^(z>>27))*0x94d049bb133111ebULL; return z^(z>>31);}
int main(int argc,char**argv){ long n=atol(argv[1]); long bad=0;
  #pragma omp parallel for reduction(+:bad)
  for(long t=0;t<64;t++){ uint64_t s=99+t*7919;
    for(long i=0;i<n/64;i++){ uint64_t r=sm(&s); float y,x; uint32_t a=(uint32_t)r,b=(uint32_t)(r>>32); int mode=i&3;
      if(mode==0){ y=urh_u2f(a); x=urh_u2f(b);} else if(mode==1){ y=((int32_t)a)/2147483648.0f; x=((int32_t)b)/2147483648.0f;}
      else if(mode==2){ y=((int32_t)a)/2147483648.0f*0.3f; x=0.9f+((int32_t)b)/2147483648.0f*0.2f;}
      else { y=urh_u2f((a&0x807fffffu)|(((a>>23)%%40+107)<<23)); x=urh_u2f((b&0x807fffffu)|(((b>>23)%%40+107)<<23)); }
      float r1=atan2f(y,x), r2=urh_atan2f(y,x); if(memcmp(&r1,&r2,4)!=0 && !(r1!=r1 && r2!=r2)) bad++;
      float q1=atanf(y), q2=urh_atanf(y); if(memcmp(&q1,&q2,4)!=0 && !(q1!=q1 && q2!=q2)) bad++; }}
  printf("%%ld\n",bad); return 0; }
'''


def test_port_equals_libm():
    hdr = os.path.join(ROOT, "urh_amd", "csrc", "fdlibm_atan2f.h")
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "chk.c")
        open(c, "w").write(SRC % hdr)
        exe = os.path.join(d, "chk")
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", c, "-o", exe, "-lm"])
        out = subprocess.check_output([exe, "120000000"]).decode().strip()
    assert out == "0", f"{out} mismatches against libm"
