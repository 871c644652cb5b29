#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
static inline float divu(float a, float b, float y){
  float aa=fabsf(a);
  if(!(aa>1.0e-25f && aa<1.0e25f)) return a/b;
  float q0=a*y; float r0=fmaf(-b,q0,a); float q1=fmaf(r0,y,q0); float r1=fmaf(-b,q1,a); return fmaf(r1,y,q1);
}
int main(){
  float bs[]={640.f,480.f,320.f,240.f,40.f,30.f,64.f,48.f,0.03f,0.36f,0.029999999f,0.08f,0.12f,3.f,7.f,1024.f,768.f,1281.f,0.0075f,1.9999999f,1.0000001f,16777215.f,0.33333334f};
  int nb=sizeof(bs)/sizeof(bs[0]);
  for(int bi=0;bi<nb;bi++){
    float b=bs[bi]; float y=1.0f/b; unsigned long long bad=0;
    #pragma omp parallel for reduction(+:bad) schedule(static)
    for(long long i=0;i<(1LL<<32);i++){
      uint32_t u=(uint32_t)i; float a; memcpy(&a,&u,4);
      float q=divu(a,b,y), r=a/b;
      uint32_t uq,ur; memcpy(&uq,&q,4); memcpy(&ur,&r,4);
      if(uq!=ur && !(q!=q && r!=r)) { if(!(q==0 && r==0)) bad++; }
    }
    printf("b=%.9g mismatches (excluding NaN payload / signed zero) = %llu\n", b, bad);
    fflush(stdout);
  }
  /* random b, random a */
  unsigned long long bad=0; uint64_t st=88172645463325252ULL;
  for(long long t=0;t<400000000LL;t++){
    st^=st<<13; st^=st>>7; st^=st<<17; uint32_t ub=(uint32_t)(st>>32)&0x7fffffffu, ua=(uint32_t)st;
    float a,b; memcpy(&a,&ua,4); memcpy(&b,&ub,4);
    if(!(b>1e-10f && b<1e10f)) continue;
    float q=divu(a,b,1.0f/b), r=a/b; uint32_t uq,ur; memcpy(&uq,&q,4); memcpy(&ur,&r,4);
    if(uq!=ur && !(q!=q&&r!=r) && !(q==0&&r==0)) bad++;
  }
  printf("random (a,b): mismatches = %llu\n", bad);
}
