#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
static float u2f(uint32_t u){float f;memcpy(&f,&u,4);return f;}
static uint32_t f2u(float f){uint32_t u;memcpy(&u,&f,4);return u;}
static float psin(float r,int k){
    const float z=r*r; float p=2.6052944122056942e-06f;
    p=fmaf(p,z,-0.00019809351942967623f); p=fmaf(p,z,0.008333061821758747f); p=fmaf(p,z,-0.16666659712791443f);
    const float v=fmaf(r*z,p,r);
    return u2f(f2u(v)^((uint32_t)(k&1)<<31));
}
static const double INV_PI=3.18309886183790671538e-01, PI_HI=3.14159265358979311600e+00, PI_LO=1.22464679914735317723e-16;
static float my_sin(float x){
    if(!(fabsf(x)<1.0e9f)) return sinf(x);
    const double xd=x, kd=rint(xd*INV_PI);
    double r=fma(-kd,PI_HI,xd); r=fma(-kd,PI_LO,r);
    const float v=psin((float)r,(int)kd);
    return fabsf(x)<0x1p-13f?x:v;
}
static float my_cos(float x){
    if(!(fabsf(x)<1.0e9f)) return cosf(x);
    const double xd=x, kd=rint(fma(xd,INV_PI,-0.5)), md=kd+0.5;
    double r=fma(-md,PI_HI,xd); r=fma(-md,PI_LO,r);
    return psin((float)r,(int)kd+1);
}
static double maxulp=0,maxrel=0; static float worst=0; static int worstfn=0;
static void chk(float x){
    for(int fn=0;fn<2;++fn){
        const double ref=fn?cos((double)x):sin((double)x);
        const float got=fn?my_cos(x):my_sin(x);
        const float rf=(float)ref;
        double ulp=fabs((double)nextafterf(fabsf(rf),INFINITY)-fabs((double)rf)); if(ulp==0)ulp=1e-45;
        const double e=fabs((double)got-ref)/ulp, rel=ref!=0?fabs((double)got-ref)/fabs(ref):(got==0?0:1);
        if(e>maxulp){maxulp=e;worst=x;worstfn=fn;} if(rel>maxrel)maxrel=rel;
    }
}
int main(){
    srand48(5);
    for(long i=0;i<30000000;++i){ chk((float)((drand48()-0.5)*20)); chk((float)((drand48()-0.5)*2e5)); chk((float)((drand48()-0.5)*2e9)); chk((float)((drand48()-0.5)*1e-3)); }
    printf("random: max ulp %.3f (x=%.9g fn=%d) max rel %.3g\n",maxulp,worst,worstfn,maxrel);
    maxulp=maxrel=0;
    for(long k=-2000000;k<=2000000;++k){ for(int h=0;h<2;++h){ float c=(float)((k+0.5*h)*M_PI); float x=c; for(int s=0;s<4;++s){chk(x);x=nextafterf(x,INFINITY);} x=c; for(int s=0;s<4;++s){chk(x);x=nextafterf(x,-INFINITY);} } }
    printf("near zeros: max ulp %.3f (x=%.9g fn=%d) max rel %.3g\n",maxulp,worst,worstfn,maxrel);
    printf("sin(-0)=%g signbit %d; cos(0)=%g; sin(1e-20)=%g sin(inf)=%g sin(1e30f)=%g vs %g\n", my_sin(-0.0f), signbit(my_sin(-0.0f)), my_cos(0.0f), my_sin(1e-20f), my_sin(INFINITY), my_sin(1e30f), sinf(1e30f));
    return 0;
}
