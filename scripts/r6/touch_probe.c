#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/time.h>
#include <sys/mman.h>
static double now(){struct timeval tv;gettimeofday(&tv,0);return tv.tv_sec+1e-6*tv.tv_usec;}
int main(int argc,char**argv){ size_t gb=atoi(argv[1]); int huge=atoi(argv[2]);
 for(int rep=0;rep<3;rep++){ double t=now(); char*p=mmap(0,gb<<30,PROT_READ|PROT_WRITE,MAP_PRIVATE|MAP_ANONYMOUS,-1,0); if(huge) madvise(p,gb<<30,MADV_HUGEPAGE);
 #pragma omp parallel for
 for(size_t i=0;i<gb*1024;i++) memset(p+(i<<20),1,1<<20);
 double t1=now();
 #pragma omp parallel for
 for(size_t i=0;i<gb*1024;i++) memset(p+(i<<20),2,1<<20);
 double t2=now(); munmap(p,gb<<30); fprintf(stderr,"rep %d huge %d: first touch %.2f s, second pass %.2f s, munmap %.2f s\n",rep,huge,t1-t,t2-t1,now()-t2);} _exit(0);}
