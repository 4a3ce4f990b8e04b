// Microbenchmark: streaming READ bandwidth of (int col[], double val[]) pairs at different per-lane widths,
// with and without an 8-byte gather x[col]. Dev aid for the SpMV design (not product code).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(e) do{hipError_t _e=(e); if(_e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1);} }while(0)

// V1: lane <-> one entry (4B + 8B per lane), NI entries per thread strided by 256
template<int NI, bool GATHER>
__global__ __launch_bounds__(256) void k_narrow(const int* __restrict__ col, const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ out, long nnz) {
    long base = (long)blockIdx.x * 256 * NI;
    int cc[NI]; double vv[NI];
    #pragma unroll
    for (int i=0;i<NI;++i){ long j = base + threadIdx.x + i*256; if (j<nnz){ cc[i]=col[j]; vv[i]=val[j]; } else {cc[i]=0; vv[i]=0;} }
    double s=0;
    #pragma unroll
    for (int i=0;i<NI;++i){ s += GATHER ? vv[i]*x[cc[i]] : vv[i]*(double)cc[i]; }
    if (s == 1.2345e-300) out[blockIdx.x*256+threadIdx.x]=s;
}
// V2: lane <-> 4 consecutive entries (16B col + 2x16B val)
template<int NI, bool GATHER>
__global__ __launch_bounds__(256) void k_wide(const int* __restrict__ col, const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ out, long nnz) {
    long base = (long)blockIdx.x * 1024 * NI;
    int4 cc[NI]; double2 va[NI], vb[NI];
    #pragma unroll
    for (int i=0;i<NI;++i){ long j = base + 4*threadIdx.x + i*1024; if (j+3<nnz){ cc[i]=*(const int4*)(col+j); va[i]=*(const double2*)(val+j); vb[i]=*(const double2*)(val+j+2);} else {cc[i]=make_int4(0,0,0,0); va[i]=vb[i]=make_double2(0,0);} }
    double s=0;
    #pragma unroll
    for (int i=0;i<NI;++i){
        if (GATHER) s += va[i].x*x[cc[i].x] + va[i].y*x[cc[i].y] + vb[i].x*x[cc[i].z] + vb[i].y*x[cc[i].w];
        else s += va[i].x*cc[i].x + va[i].y*cc[i].y + vb[i].x*cc[i].z + vb[i].y*cc[i].w;
    }
    if (s == 1.2345e-300) out[blockIdx.x*256+threadIdx.x]=s;
}

// V3 family: the lean SpMV built up step by step. STEP 0: load+gather+sum (no LDS); 1: + write y (1 double / thread);
// 2: products -> LDS, sync, thread-per-row reduce from LDS, write y; REMAP: XCD-contiguous block order
template<int STEP, bool REMAP>
__global__ __launch_bounds__(256) void k_lean(const int* __restrict__ crow, const int* __restrict__ col, const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y, int n, int nblk) {
    constexpr int NI=5;
    __shared__ double prod[1280];
    __shared__ int crowL[257];
    const int t = threadIdx.x;
    int b = blockIdx.x;
    if (REMAP) { const int per=(nblk+7)>>3; b = (blockIdx.x&7)*per + (blockIdx.x>>3); if (b>=nblk) return; }
    const int r0 = b*256; const int nr = (n-r0<256)?n-r0:256;
    if (t<=nr) crowL[t]=crow[r0+t];
    if (t==0 && nr==256) crowL[256]=crow[r0+256];
    __syncthreads();
    const int j0=crowL[0]; const int cnt=crowL[nr]-j0;
    int cc[NI]; double vv[NI];
    #pragma unroll
    for(int i=0;i<NI;++i){ int j=t+i*256; if(j<cnt){cc[i]=col[j0+j]; vv[i]=val[j0+j];} }
    if (STEP<=1) {
        double s=0;
        #pragma unroll
        for(int i=0;i<NI;++i){ int j=t+i*256; if(j<cnt) s+=vv[i]*x[cc[i]]; }
        if (STEP==1) { if (t<nr) y[r0+t]=s; } else if (s==1.2345e-300) y[r0+t]=s;
        return;
    }
    #pragma unroll
    for(int i=0;i<NI;++i){ int j=t+i*256; if(j<cnt) prod[j]=vv[i]*x[cc[i]]; }
    __syncthreads();
    if (t<nr){ int lo=crowL[t]-j0, hi=crowL[t+1]-j0; double s=0; for(int j=lo;j<hi;++j) s+=prod[j]; y[r0+t]=s; }
}


// L family: lean SpMV with the crow->col dependency broken: the block's entry range comes from two scalar
// loads crow[r0], crow[r0+nr]; per-row pointers are loaded in parallel with col/val.
// ROWS = rows per block (256 or 512), THREADS = 256 or 512, DOT: fused tile partial <w,y>
template<int ROWS, int THREADS, bool DOT>
__global__ __launch_bounds__(THREADS) void k_lean2(const int* __restrict__ crow, const int* __restrict__ col, const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y, const double* __restrict__ w, double* __restrict__ tpart, int n, int nblk) {
    constexpr int CAP = ROWS*5;
    constexpr int NI = CAP/THREADS;
    constexpr int RPT = ROWS/THREADS;
    __shared__ double prod[CAP];
    __shared__ int crowL[ROWS+1];
    __shared__ double red[THREADS];
    const int t = threadIdx.x;
    const int per=(nblk+7)>>3; const int b = (blockIdx.x&7)*per + (blockIdx.x>>3); if (b>=nblk) return;
    const int r0 = b*ROWS; const int nr = (n-r0<ROWS)?n-r0:ROWS;
    const int j0 = __builtin_amdgcn_readfirstlane(crow[r0]);
    const int j1 = __builtin_amdgcn_readfirstlane(crow[r0+nr]);
    const int cnt = j1-j0;
    int cc[NI]; double vv[NI];
    #pragma unroll
    for(int i=0;i<NI;++i){ int j=t+i*THREADS; if(j<cnt){cc[i]=col[j0+j]; vv[i]=val[j0+j];} }
    for (int i=t;i<=nr;i+=THREADS) crowL[i]=crow[r0+i];
    #pragma unroll
    for(int i=0;i<NI;++i){ int j=t+i*THREADS; if(j<cnt) prod[j]=vv[i]*x[cc[i]]; }
    __syncthreads();
    double acc=0;
    #pragma unroll
    for (int k=0;k<RPT;++k){ int r=t+k*THREADS; if (r<nr){ int lo=crowL[r]-j0, hi=crowL[r+1]-j0; double s=0; for(int j=lo;j<hi;++j) s+=prod[j]; y[r0+r]=s; if (DOT) acc=fma(w[r0+r],s,acc);} }
    if (DOT) {
        red[t]=acc; __syncthreads();
        for (int s2=THREADS/2; s2>=64; s2>>=1){ if (t<s2) red[t]+=red[t+s2]; __syncthreads(); }
        if (t<64){ double a=red[t]; if (THREADS==64) a=acc; for(int o=32;o>=1;o>>=1) a+=__shfl_down(a,o); if(t==0) tpart[b]=a; }
    }
}


// N family: lean step2 with non-temporal loads on the pure streaming arrays so the Infinity Cache keeps x/y/crow.
// NT bit0: val, bit1: col, bit2: crow, bit3: y store nt
template<int NT>
__global__ __launch_bounds__(256) void k_lean_nt(const int* __restrict__ crow, const int* __restrict__ col, const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y, int n, int nblk) {
    constexpr int NI=5;
    __shared__ double prod[1280];
    __shared__ int crowL[257];
    const int t = threadIdx.x;
    const int per=(nblk+7)>>3; const int b = (blockIdx.x&7)*per + (blockIdx.x>>3); if (b>=nblk) return;
    const int r0 = b*256; const int nr = (n-r0<256)?n-r0:256;
    if (t<=nr) crowL[t]= (NT&4) ? __builtin_nontemporal_load(crow+r0+t) : crow[r0+t];
    if (t==0 && nr==256) crowL[256]=crow[r0+256];
    __syncthreads();
    const int j0=crowL[0]; const int cnt=crowL[nr]-j0;
    int cc[NI]; double vv[NI];
    #pragma unroll
    for(int i=0;i<NI;++i){ int j=t+i*256; if(j<cnt){cc[i]=(NT&2)?__builtin_nontemporal_load(col+j0+j):col[j0+j]; vv[i]=(NT&1)?__builtin_nontemporal_load(val+j0+j):val[j0+j];} }
    #pragma unroll
    for(int i=0;i<NI;++i){ int j=t+i*256; if(j<cnt) prod[j]=vv[i]*x[cc[i]]; }
    __syncthreads();
    if (t<nr){ int lo=crowL[t]-j0, hi=crowL[t+1]-j0; double s=0; for(int j=lo;j<hi;++j) s+=prod[j]; if (NT&8) __builtin_nontemporal_store(s, y+r0+t); else y[r0+t]=s; }
}

template<typename F> float timeit(F f, int reps){ hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); for(int i=0;i<3;++i) f(); CK(hipEventRecord(a)); for(int i=0;i<reps;++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms,a,b)); return ms/reps; }
int main(int argc, char** argv){
    const int nx = 2000; const long n = (long)nx*nx; 
    std::vector<int> hc; std::vector<double> hv; hc.reserve(5*n); hv.reserve(5*n);
    for (long k=0;k<n;++k){ long i=k/nx, j=k%nx; if(i>0){hc.push_back(k-nx);hv.push_back(-1);} if(j>0){hc.push_back(k-1);hv.push_back(-1);} hc.push_back(k);hv.push_back(4); if(j<nx-1){hc.push_back(k+1);hv.push_back(-1);} if(i<nx-1){hc.push_back(k+nx);hv.push_back(-1);} }
    long nnz = hc.size(); printf("n=%ld nnz=%ld\n", n, nnz);
    int* col; double *val,*x,*out; CK(hipMalloc(&col,nnz*4+64)); CK(hipMalloc(&val,nnz*8+64)); CK(hipMalloc(&x,n*8)); CK(hipMalloc(&out,1<<24));
    CK(hipMemcpy(col,hc.data(),nnz*4,hipMemcpyHostToDevice)); CK(hipMemcpy(val,hv.data(),nnz*8,hipMemcpyHostToDevice)); CK(hipMemset(x,0,n*8));
    double bytes = nnz*12.0, bytes_g = nnz*12.0 + n*8.0;
    #define RUN(name, kern, per, by) { int grid=(int)((nnz + (per)-1)/(per)); float ms=timeit([&]{ kern<<<grid,256>>>(col,val,x,out,nnz); },50); printf("%-28s %8.1f us  %7.1f GB/s\n", name, ms*1e3, by/ms/1e6); }
    RUN("narrow NI=5  stream", (k_narrow<5,false>), 256*5, bytes);
    RUN("narrow NI=10 stream", (k_narrow<10,false>), 256*10, bytes);
    RUN("wide   NI=1  stream", (k_wide<1,false>), 1024*1, bytes);
    RUN("wide   NI=2  stream", (k_wide<2,false>), 1024*2, bytes);
    RUN("wide   NI=4  stream", (k_wide<4,false>), 1024*4, bytes);
    RUN("narrow NI=5  +gather", (k_narrow<5,true>), 256*5, bytes_g);
    RUN("narrow NI=10 +gather", (k_narrow<10,true>), 256*10, bytes_g);
    RUN("wide   NI=1  +gather", (k_wide<1,true>), 1024*1, bytes_g);
    RUN("wide   NI=2  +gather", (k_wide<2,true>), 1024*2, bytes_g);
    RUN("wide   NI=4  +gather", (k_wide<4,true>), 1024*4, bytes_g);
    std::vector<int> hr(n+1); { long p=0; for(long k=0;k<n;++k){ hr[k]=(int)p; long i=k/nx,j=k%nx; p+=1+(i>0)+(j>0)+(j<nx-1)+(i<nx-1);} hr[n]=(int)p; }
    int* crow; double* y; CK(hipMalloc(&crow,(n+1)*4)); CK(hipMalloc(&y,n*8)); CK(hipMemcpy(crow,hr.data(),(n+1)*4,hipMemcpyHostToDevice));
    int nblk=(int)((n+255)/256); int gridr=((nblk+7)/8)*8; double by=nnz*12.0+(n+1)*4.0+n*16.0;
    #define RUNL(name, S, R) { float ms=timeit([&]{ k_lean<S,R><<<(R?gridr:nblk),256>>>(crow,col,val,x,y,(int)n,nblk); },50); printf("%-36s %8.1f us  %7.1f GB/s (alg bytes)\n", name, ms*1e3, by/ms/1e6); }
    RUNL("lean step0 (no y, no LDS)", 0, false);
    RUNL("lean step1 (+y write)", 1, false);
    RUNL("lean step2 (LDS reduce) ", 2, false);
    RUNL("lean step0 remap", 0, true);
    RUNL("lean step1 remap", 1, true);
    RUNL("lean step2 remap", 2, true);
    double* w; double* tp; CK(hipMalloc(&w,n*8)); CK(hipMemset(w,0,n*8)); CK(hipMalloc(&tp,1<<20));
    #define RUNL2(name, ROWS, TH, DOT) { int nb=(int)((n+ROWS-1)/ROWS); int gr=((nb+7)/8)*8; float ms=timeit([&]{ k_lean2<ROWS,TH,DOT><<<gr,TH>>>(crow,col,val,x,y,w,tp,(int)n,nb); },50); printf("%-36s %8.1f us  %7.1f GB/s (alg bytes)\n", name, ms*1e3, by/ms/1e6); }
    RUNL2("lean2 256r/256t", 256, 256, false);
    RUNL2("lean2 512r/256t", 512, 256, false);
    RUNL2("lean2 512r/512t", 512, 512, false);
    RUNL2("lean2 1024r/1024t", 1024, 1024, false);
    RUNL2("lean2 128r/128t", 128, 128, false);
    RUNL2("lean2 256r/256t +dot", 256, 256, true);
    RUNL2("lean2 512r/512t +dot", 512, 512, true);
    #define RUNN(name, NT) { float ms=timeit([&]{ k_lean_nt<NT><<<gridr,256>>>(crow,col,val,x,y,(int)n,nblk); },50); printf("%-36s %8.1f us  %7.1f GB/s (alg bytes)\n", name, ms*1e3, by/ms/1e6); }
    RUNN("lean nt=0", 0);
    RUNN("lean nt val", 1);
    RUNN("lean nt val+col", 3);
    RUNN("lean nt val+col+crow", 7);
    RUNN("lean nt val+col+crow+ystore", 15);
    RUNN("lean nt val+col +ystore", 11);
    CK(hipDeviceSynchronize());
    return 0;
}
