#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <omp.h>
#include "host_pack.h"
static double now(){struct timespec t;clock_gettime(CLOCK_MONOTONIC,&t);return t.tv_sec*1e3+t.tv_nsec*1e-6;}
int main(int argc,char**argv){
  size_t n=argc>1?atol(argv[1]):200000; int L=1000, stride=1028; int nt=argc>2?atoi(argv[2]):8;
  char*buf=malloc(n*2*stride); const char*al="ACGT";
  unsigned s=1; for(size_t i=0;i<n*2*stride;i++){ s=s*1664525u+1013904223u; buf[i]=((i%stride)<L)?al[s>>30]:0; }
  size_t wps=(L+15)/16+1; uint32_t*out=malloc(n*2*wps*4), *out2=malloc(n*2*wps*4); memset(out,1,n*2*wps*4); memset(out2,2,n*2*wps*4);
  omp_set_num_threads(nt);
  for(int rep=0;rep<4;rep++){
    double t0=now(); int bad=0;
    #pragma omp parallel for reduction(|:bad) schedule(static)
    for(size_t i=0;i<2*n;i++) bad|=wfagpu_host_pack_sequence(buf+i*stride,L,out+i*wps);
    double t1=now();
    printf("avx2 %d threads: %.2f ms  %.1f GB/s ascii  bad=%d\n",nt,t1-t0,n*2.0*stride/(t1-t0)/1e6,bad);
  }
  { char*cp=malloc(n*2*stride); memset(cp,0,n*2*stride);
    for(int rep=0;rep<3;rep++){ double t0=now();
      #pragma omp parallel for schedule(static)
      for(size_t i=0;i<2*n;i++) memcpy(cp+i*stride,buf+i*stride,stride);
      double t1=now(); printf("memcpy %d threads: %.2f ms %.1f GB/s\n",nt,t1-t0,n*2.0*stride/(t1-t0)/1e6);} free(cp);}
  double t0=now(); int bad=0;
  #pragma omp parallel for reduction(|:bad) schedule(static)
  for(size_t i=0;i<2*n;i++) bad|=wfagpu_host_pack_sequence_scalar(buf+i*stride,L,out2+i*wps);
  double t1=now();
  printf("scalar: %.2f ms %.1f GB/s  same=%d\n",t1-t0,n*2.0*stride/(t1-t0)/1e6, memcmp(out,out2,n*2*wps*4)==0);
  return 0; }
