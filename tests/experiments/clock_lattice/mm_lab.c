// experiment helper: M&M chains from arbitrary start states (same arithmetic as oracle/xrit_oracle.c)
#include <math.h>
#include <stdint.h>
#include <string.h>
typedef struct { float re, im; } cf;
typedef struct { int64_t ii; float mu, omega; cf p0, p1, c0, c1; } st_t;
typedef struct { float omega_mid, omega_lim, gain_omega, gain_mu; } par_t;
static inline float clipf(float x, float c){ return 0.5f*(fabsf(x+c)-fabsf(x-c)); }
static inline cf step(const cf*x, const float*table, st_t*s, const par_t*p, int*arm){
    cf p2=s->p1, p1=s->p0, c2=s->c1, c1=s->c0;
    int imu=(int)rintf(s->mu*128.f);
    if(arm)*arm=imu;
    const float*row=table+imu*8;
    float ar=0, ai=0;
    const cf*w=x+s->ii;
    for(int k=0;k<8;k++){ ar+=row[7-k]*w[k].re; ai+=row[7-k]*w[k].im; }
    cf p0={ar,ai};
    cf c0={ar>0?1.f:0.f, ai>0?1.f:0.f};
    float dcr=c0.re-c2.re, dci=c0.im-c2.im;
    float xr=dcr*p1.re+dci*p1.im;
    float dpr=p0.re-p2.re, dpi=p0.im-p2.im;
    float yr=dpr*c1.re+dpi*c1.im;
    float mm=clipf(yr-xr,1.f);
    float omega=s->omega+p->gain_omega*mm;
    omega=p->omega_mid+clipf(omega-p->omega_mid,p->omega_lim);
    float mu=s->mu+omega+p->gain_mu*mm;
    float fl=floorf(mu);
    s->ii+=(int64_t)fl; s->mu=mu-fl; s->omega=omega;
    s->p1=p1; s->p0=p0; s->c1=c1; s->c0=c0;
    return p0;
}
// run K chains of ns symbols each. S,E arrays. out (optional): K*ns complex; trace arrays optional (mu, omega, arm)
void run_chains(const cf*x, int64_t ni, const float*table, const par_t*p, const st_t*S, st_t*E, int K, int ns,
                cf*out, float*mu_tr, float*om_tr, int*arm_tr, int*produced)
{
    #pragma omp parallel for schedule(static)
    for(int k=0;k<K;k++){
        st_t s=S[k]; int n=0;
        for(int i=0;i<ns;i++){
            if(s.ii>=ni||s.ii<0) break;
            int arm; 
            if(mu_tr) mu_tr[(int64_t)k*ns+i]=s.mu;
            if(om_tr) om_tr[(int64_t)k*ns+i]=s.omega;
            cf q=step(x,table,&s,p,&arm);
            if(arm_tr) arm_tr[(int64_t)k*ns+i]=arm;
            if(out) out[(int64_t)k*ns+i]=q;
            n++;
        }
        E[k]=s; if(produced) produced[k]=n;
    }
}
