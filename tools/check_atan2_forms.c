// The PLL's branch-free approx_atan2 (AudioSDR.h:384-408 as asdr_kernels.hip evaluates it): the round-4 form (quadrant selects as the reference
// writes them) against the round-5 form (sign-bit half_pi, no x == 0 branch of its own), bit for bit, NaNs included, over special values and N
// random operand pairs.   gcc -O2 -ffp-contract=off -msse2 -mfpmath=sse tools/check_atan2_forms.c -lm -o /tmp/check_atan2 && /tmp/check_atan2 [N]
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
static const double PI_D = 3.14159265358979323846;
static float approx_atan(float z){ const float n1=0.97239411f,n2=-0.19194795f; return (n1+n2*z*z)*z; }
static float old_f(float y,float x,float half_pi){
  const int xnz=(x!=0.0f); const int big=fabsf(x)>fabsf(y);
  const float num=big?y:x, den=big?x:y; const float z=num/den; const float a=approx_atan(z);
  const double pis=(y>=0.0f)?PI_D:-PI_D; const float a_pi=(float)((double)a+pis);
  const float r_big=(x>0.0f)?a:a_pi; const float r_small=-a+((y>0.0f)?half_pi:-half_pi);
  const float r_x0=(y>0.0f)?half_pi:((y<0.0f)?-half_pi:0.0f);
  return xnz?(big?r_big:r_small):r_x0; }
static float new_f(float y,float x,float half_pi){
  const int big=fabsf(x)>fabsf(y);
  const float num=big?y:x, den=big?x:y; const float z=num/den; const float a=approx_atan(z);
  const double pis=(y>=0.0f)?PI_D:-PI_D; const float a_pi=(float)((double)a+pis);
  const float r_big=(x>0.0f)?a:a_pi; const float r_small=-a+copysignf(half_pi,y);
  const float r=big?r_big:r_small; return (num==0.0f&&!(fabsf(den)>0.0f))?0.0f:r; }
static uint32_t bits(float f){uint32_t u;memcpy(&u,&f,4);return u;}
static float fb(uint32_t u){float f;memcpy(&f,&u,4);return f;}
int main(int argc,char**argv){ long N = argc>1 ? atol(argv[1]) : 400000000L;
  const float hp=(float)(PI_D/2.0);
  uint32_t sp[]={0,0x80000000u,1,0x80000001u,0x007fffffu,0x00800000u,0x3f800000u,0xbf800000u,0x7f7fffffu,0xff7fffffu,0x7f800000u,0xff800000u,0x7fc00000u,0xffc00000u,0x7f800001u,0x3eaaaaabu,0x34000000u,0xb4000000u};
  int ns=sizeof sp/4; long bad=0, n=0, nanmis=0;
  for(int i=0;i<ns;i++)for(int j=0;j<ns;j++){float y=fb(sp[i]),x=fb(sp[j]);float o=old_f(y,x,hp),w=new_f(y,x,hp);n++;
    if(bits(o)!=bits(w)){ if(isnan(o)&&isnan(w)) nanmis++; else {bad++; printf("special y=%08x x=%08x old=%08x new=%08x\n",sp[i],sp[j],bits(o),bits(w));}}}
  uint64_t s=88172645463325252ull;
  for(long k=0;k<N;k++){ s^=s<<13; s^=s>>7; s^=s<<17; uint32_t a=(uint32_t)s, b=(uint32_t)(s>>32);
    if((k&7)==0) b=(b&0x80000000u); if((k&7)==1) a=(a&0x80000000u); if((k&15)==2) b=a^(b&0x80000001u);
    float y=fb(a),x=fb(b); float o=old_f(y,x,hp),w=new_f(y,x,hp); n++;
    if(bits(o)!=bits(w)){ if(isnan(o)&&isnan(w)) nanmis++; else { if(bad<10) printf("y=%08x x=%08x old=%08x new=%08x\n",a,b,bits(o),bits(w)); bad++; } } }
  printf("checked %ld, non-NaN mismatches %ld, NaN-bit mismatches %ld\n",n,bad,nanmis); return bad!=0; }
