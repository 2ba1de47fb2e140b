// CPU check of the exact integer colour path against the reference formula (IEEE f64), full 10-bit cube.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
static uint8_t clampu8(double x){ if (x<0.) return 0; if (x>255.) return 255; return (uint8_t)x; }
static void ref(uint16_t y16,uint16_t u16,uint16_t v16,uint8_t*o){ double y=y16,u=u16,v=v16;
  double r=y+1.57480*(v-512.), g=y-0.18733*(u-512.)-(0.46813*(v-512.)), b=y+1.85563*(u-512.);
  o[0]=clampu8(floor(r/1023.*255.)); o[1]=clampu8(floor(g/1023.*255.)); o[2]=clampu8(floor(b/1023.*255.)); }
// returns 0..255, or -1 when the quotient is an exact integer (caller must take the exact f64 path)
static int chan(int64_t n17, uint32_t d){ if (n17 <= 0) return 0; uint32_t x=(uint32_t)n17; uint32_t q=x/d; if (q*d==x) return -1; return q>255?255:(int)q; }
int main(){ uint64_t bad=0, slow=0, total=0; 
  for (int y=0;y<1024;++y) for (int u=0;u<1024;++u) for (int v=0;v<1024;++v){ int du=u-512, dv=v-512;
    int64_t nr=17ll*(2500ll*y+3937ll*dv), ng=17ll*(100000ll*y-18733ll*du-46813ll*dv), nb=17ll*(100000ll*y+185563ll*du);
    if (nr>4294967295ll||ng>4294967295ll||nb>4294967295ll){printf("overflow\n");return 1;}
    int r=chan(nr,170500u), g=chan(ng,6820000u), b=chan(nb,6820000u); uint8_t o[3]; ref(y,u,v,o); total++;
    if (r<0||g<0||b<0){ slow++; continue; }
    if (r!=o[0]||g!=o[1]||b!=o[2]) { if (bad<5) printf("mismatch %d %d %d: %d %d %d vs %d %d %d\n",y,u,v,r,g,b,o[0],o[1],o[2]); bad++; } }
  printf("total %llu mismatches %llu exact-multiple(slow path) %llu\n",(unsigned long long)total,(unsigned long long)bad,(unsigned long long)slow); return bad!=0; }
