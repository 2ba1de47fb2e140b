// CPU check of the kernels' colour evaluation (tmc2-rs_amd/csrc/vpcc_colour.h, compiled here from the same
// source) against the reference formula (IEEE f64, src/codec.rs:661-687) on the full 10-bit cube: every
// triplet is either flagged ambiguous (the kernel then evaluates the reference formula) or equal.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include "vpcc_colour.h"
static uint8_t clampu8(double x){ if (x<0.) return 0; if (x>255.) return 255; return (uint8_t)x; }
static void ref(uint16_t y16,uint16_t u16,uint16_t v16,uint8_t*o){ double y=y16,u=u16,v=v16;
  double r=y+1.57480*(v-512.), g=y-0.18733*(u-512.)-(0.46813*(v-512.)), b=y+1.85563*(u-512.);
  o[0]=clampu8(floor(r/1023.*255.)); o[1]=clampu8(floor(g/1023.*255.)); o[2]=clampu8(floor(b/1023.*255.)); }
int main(){ uint64_t bad=0, slow=0, total=0, exact_unflagged=0;
  for (int u=0;u<1024;++u) for (int v=0;v<1024;++v){ const vpcc_chroma_part c = vpcc_colour_chroma(u, v);
    const int du=u-512, dv=v-512;
    for (int y=0;y<1024;++y){
      uint32_t fmin=0xFFFFFFFFu; const uint32_t rgb=vpcc_colour_luma(y,c,&fmin); const int amb = fmin==0; uint8_t o[3]; ref(y,u,v,o); total++;
      { uint32_t f1=0xFFFFFFFFu; const uint32_t one=vpcc_colour_one(y,u,v,VPCC_K_R,VPCC_K_G,VPCC_K_B,&f1); if (one!=rgb || (f1==0)!=amb) { if (bad<5) printf("vpcc_colour_one differs at %d %d %d\n",y,u,v); bad++; } }
      const int64_t nr=17ll*(2500ll*y+3937ll*dv), ng=17ll*(100000ll*y-18733ll*du-46813ll*dv), nb=17ll*(100000ll*y+185563ll*du);
      const int exact = (nr>0&&nr%170500==0)||(ng>0&&ng%6820000==0)||(nb>0&&nb%6820000==0);
      if (amb){ slow++; continue; }
      if (exact) exact_unflagged++;
      if ((rgb&255)!=o[0]||((rgb>>8)&255)!=o[1]||(rgb>>16)!=o[2]) { if (bad<5) printf("mismatch %d %d %d: %06x vs %d %d %d\n",y,u,v,rgb,o[0],o[1],o[2]); bad++; } } }
  printf("total %llu mismatches %llu ambiguous(slow path) %llu exact-unflagged %llu\n",(unsigned long long)total,(unsigned long long)bad,(unsigned long long)slow,(unsigned long long)exact_unflagged); return bad!=0 || exact_unflagged!=0; }
