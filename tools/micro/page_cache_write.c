// Page-cache write rate of this host: T threads pwrite (or memcpy into a shared mapping) disjoint chunks of one file or of T files.
// What bounds the routing pass of files mode (mnc_fastq_route: a gigabyte appended to mapped/<sample>.fastq per 100 000 reads).
// gcc -O2 -pthread -o /tmp/pcw tools/micro/page_cache_write.c; /tmp/pcw <threads> <files> <0 pwrite | 1 mmap | 2 mmap, MADV_POPULATE_WRITE per chunk | 3 mmap over fallocated blocks> <MB> <chunk KB> <dir> [pre = ftruncate first]
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <unistd.h>
#include <pthread.h>
#include <time.h>
#include <sys/mman.h>
#include <sys/stat.h>
static int T, nfiles, mode; static size_t total, chunk; static int fds[64]; static char *map[64];
static double now(){struct timespec t; clock_gettime(CLOCK_MONOTONIC,&t); return t.tv_sec+1e-9*t.tv_nsec;}
static void *work(void *a){ long t=(long)a; char *buf=malloc(chunk); memset(buf,'A'+t,chunk);
  size_t per=total/T; int f=nfiles==1?0:t%nfiles; off_t base=(nfiles==1? per*t : per*(t/nfiles));
  for(size_t o=0;o<per;o+=chunk){ memset(buf,'a'+(o/chunk)%26,chunk);
    if(mode==0){ if(pwrite(fds[f],buf,chunk,base+o)!=(ssize_t)chunk){perror("pwrite");exit(1);} }
    else { if(mode==2 && madvise(map[f]+base+o,chunk,23/*MADV_POPULATE_WRITE*/)) { perror("madvise"); exit(1);} memcpy(map[f]+base+o,buf,chunk); } }
  free(buf); return 0; }
int main(int c,char**v){ T=atoi(v[1]); nfiles=atoi(v[2]); mode=atoi(v[3]); total=(size_t)atol(v[4])<<20; chunk=(size_t)atol(v[5])<<10; const char*dir=v[6];
  char p[256]; for(int i=0;i<nfiles;i++){ snprintf(p,256,"%s/wt_%d.bin",dir,i); unlink(p); fds[i]=open(p,O_RDWR|O_CREAT,0666);
    if(mode>=1||c>7){ if(ftruncate(fds[i],total)){perror("ftruncate");} if(mode==3 && posix_fallocate(fds[i],0,total)){perror("fallocate");} }
    if(mode>=1){ map[i]=mmap(0,total,PROT_READ|PROT_WRITE,MAP_SHARED,fds[i],0); if(map[i]==MAP_FAILED){perror("mmap");exit(1);} } }
  pthread_t th[64]; double t0=now(); for(long t=0;t<T;t++) pthread_create(&th[t],0,work,(void*)t); for(int t=0;t<T;t++) pthread_join(th[t],0); double t1=now();
  printf("T=%d files=%d mode=%s total=%zu MB chunk=%zu KB: %.3f s = %.2f GB/s\n",T,nfiles,mode==0?"pwrite":mode==1?"mmap":mode==2?"mmap+populate":"mmap+fallocate",total>>20,chunk>>10,t1-t0,total/(t1-t0)/1e9);
  for(int i=0;i<nfiles;i++){ snprintf(p,256,"%s/wt_%d.bin",dir,i); close(fds[i]); unlink(p);} return 0; }
