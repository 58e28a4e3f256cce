// Host priority recurrence on plain malloc memory vs the rate seen inside the pipeline (pinned).
// gcc -O3 prio_bench.c ../../lz77_amd/csrc/hoststage.c -o prio_bench.bin ; ./prio_bench.bin ps.bin N
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <time.h>
#include <sys/mman.h>
#include "../../lz77_amd/csrc/lz77x_internal.h"
static double now(void){struct timespec t;clock_gettime(CLOCK_MONOTONIC,&t);return t.tv_sec+t.tv_nsec*1e-9;}
int main(int argc,char**argv){ if(argc<3) return 2; size_t n=strtoull(argv[2],0,10); int huge = argc > 3; uint32_t *ps, *xv;
 if (huge) { posix_memalign((void**)&ps, 2<<20, n*4); posix_memalign((void**)&xv, 2<<20, n*4); madvise(ps, n*4, MADV_HUGEPAGE); madvise(xv, n*4, MADV_HUGEPAGE); } else { ps=malloc(n*4); xv=malloc(n*4); }
 FILE*f=fopen(argv[1],"rb"); if(!f||fread(ps,4,n,f)!=n) return 1; fclose(f); memset(xv,0,n*4);
 /* file holds distances P|S<<16; the recurrence wants ring cells */
 { uint32_t m=lz77x_prio_mask(4095); for(size_t i=0;i<n;i++){ uint32_t v=ps[i]; ps[i]=(((uint32_t)i+(v&0xFFFF))&m)|((((uint32_t)i+(v>>16))&m)<<16);} }
 for(int it=0;it<4;it++){ lz77x_prio_state st; lz77x_prio_init(&st,4095); double t=now(); lz77x_prio_run(&st,ps,4095,n,xv); double dt=now()-t;
  printf("%s: %.3f ns/pos (%.1f ms) transfers %lu\n",huge?"hugepage":"malloc",dt/n*1e9,dt*1e3,(unsigned long)st.transfers); lz77x_prio_free(&st);} return 0;}
