// Microbenchmark: register-resident radix-2 butterfly throughput for several 64-bit modular-multiply formulations (gfx950).
// Each thread holds 16 u64 values and runs 4 butterfly stages per iteration (32 butterflies), twiddles in VGPRs.
// Build: hipcc --offload-arch=gfx950 -O3 micro_bfly.hip -o micro_bfly
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64; typedef unsigned int u32;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

struct Mod { u64 q, qinv, nq, q2, q4; };

// ---- formulation 0: reference-exact Montgomery lazy butterfly (ring/ntt.go:155-161) ----
__device__ __forceinline__ u64 mredlazy(u64 x, u64 y, u64 q, u64 qinv){
  unsigned __int128 p = (unsigned __int128)x*y;
  u64 alo=(u64)p, ahi=(u64)(p>>64);
  u64 H = __umul64hi(alo*qinv,q);
  return ahi - H + q;
}
template<bool RED> __device__ __forceinline__ void bf_mont(u64& U, u64& V, u64 w, u64 wp, const Mod& m){
  if (RED) { if (U >= m.q4) U -= m.q4; }
  u64 r = mredlazy(V, w, m.q, m.qinv);
  V = U + m.q2 - r; U = U + r;
}
// ---- formulation 1: Shoup with approximate quotient (9 multiplies), r in [0,4q) ----
__device__ __forceinline__ u64 shoup_approx(u64 V, u64 w, u64 wp, u64 nq){
  u32 V0=(u32)V, V1=(u32)(V>>32), p0=(u32)wp, p1=(u32)(wp>>32);
  u32 a = __umulhi(V1,p0), b=__umulhi(V0,p1);
  u64 Q = (u64)V1*p1 + a;  Q += b;
  u32 Q0=(u32)Q, Q1=(u32)(Q>>32), w0=(u32)w, w1=(u32)(w>>32), n0=(u32)nq, n1=(u32)(nq>>32);
  u64 t = (u64)V0*w0;
  t = (u64)Q0*n0 + t;
  u32 hi = (u32)(t>>32);
  hi = hi + V0*w1 + V1*w0;
  hi = hi + Q0*n1 + Q1*n0;
  return ((u64)hi<<32) | (u32)t;
}
template<bool RED> __device__ __forceinline__ void bf_shoup(u64& U, u64& V, u64 w, u64 wp, const Mod& m){
  if (RED) { u64 t = U - m.q4; U = (U < m.q4) ? U : t; }
  u64 r = shoup_approx(V, w, wp, m.nq);
  V = U + m.q4 - r; U = U + r;
}
// ---- formulation 2: Shoup approx, conditional subtract through sign mask + bfi ----
template<bool RED> __device__ __forceinline__ void bf_shoup_mask(u64& U, u64& V, u64 w, u64 wp, const Mod& m){
  if (RED) {
    u64 t = U - m.q4;
    u32 mask = (u32)((int)(u32)(t>>32) >> 31);
    u32 lo = ((u32)U & mask) | ((u32)t & ~mask);
    u32 hi = ((u32)(U>>32) & mask) | ((u32)(t>>32) & ~mask);
    U = ((u64)hi<<32)|lo;
  }
  u64 r = shoup_approx(V, w, wp, m.nq);
  V = U + m.q4 - r; U = U + r;
}
// ---- formulation 3: exact Shoup (10 multiplies incl. full hi64), r in [0,2q) ----
__device__ __forceinline__ u64 shoup_exact(u64 V, u64 w, u64 wp, u64 q){
  u64 Q = __umul64hi(V, wp);
  return V*w - Q*q;
}
template<bool RED> __device__ __forceinline__ void bf_shoup_exact(u64& U, u64& V, u64 w, u64 wp, const Mod& m){
  if (RED) { u64 t = U - m.q4; U = (U < m.q4) ? U : t; }
  u64 r = shoup_exact(V, w, wp, m.q);
  V = U + m.q2 - r; U = U + r;
}

template<int F, bool RED> __device__ __forceinline__ void bf(u64& U, u64& V, u64 w, u64 wp, const Mod& m){
  if (F==0) bf_mont<RED>(U,V,w,wp,m);
  else if (F==1) bf_shoup<RED>(U,V,w,wp,m);
  else if (F==2) bf_shoup_mask<RED>(U,V,w,wp,m);
  else bf_shoup_exact<RED>(U,V,w,wp,m);
}

template<int F, bool SCALAR_TW>
__global__ void __launch_bounds__(256) kern(u64* data, const u64* tw, Mod m, int iters){
  int tid = blockIdx.x*256 + threadIdx.x;
  u64 x[16];
  #pragma unroll
  for(int k=0;k<16;++k) x[k] = data[(size_t)k*gridDim.x*256 + tid];
  u64 w[4], wp[4];
  #pragma unroll
  for(int s=0;s<4;++s){
    int idx = SCALAR_TW ? (blockIdx.x & 15) : (tid & 4095);
    w[s] = tw[2*(idx*4+s)]; wp[s] = tw[2*(idx*4+s)+1];
  }
  for(int it=0; it<iters; ++it){
    #pragma unroll
    for(int s=0;s<4;++s){
      const int h = 8>>s;
      #pragma unroll
      for(int k=0;k<16;++k){
        if ((k & h)==0){
          if (s&1) bf<F,true>(x[k], x[k+h], w[s], wp[s], m);
          else     bf<F,false>(x[k], x[k+h], w[s], wp[s], m);
        }
      }
    }
  }
  #pragma unroll
  for(int k=0;k<16;++k) data[(size_t)k*gridDim.x*256 + tid] = x[k];
}

typedef void (*kfn)(u64*, const u64*, Mod, int);
struct Entry { const char* name; kfn fn; };

int main(){
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop,0));
  int cus = prop.multiProcessorCount;
  const u64 q = 0x1fffffffffe00001ull;
  Mod m; m.q=q; m.q2=2*q; m.q4=4*q; m.nq=0-q;
  u64 qi=1, qq=q; for(int i=0;i<63;++i){ qi*=qq; qq*=qq; } m.qinv=qi;
  int maxblocks = cus*8;
  size_t n = (size_t)16*maxblocks*256;
  u64* h = (u64*)malloc(n*8); for(size_t i=0;i<n;++i) h[i] = (0x9e3779b97f4a7c15ull*(i+1)) % q;
  u64* d; CK(hipMalloc(&d,n*8)); CK(hipMemcpy(d,h,n*8,hipMemcpyHostToDevice));
  size_t nt = 4096*4*2; u64* ht=(u64*)malloc(nt*8);
  for(size_t i=0;i<nt/2;++i){ u64 w=(0xd1342543de82ef95ull*(i+7))%q; unsigned __int128 z=((unsigned __int128)w<<64)/q; ht[2*i]=w; ht[2*i+1]=(u64)z; }
  u64* dt; CK(hipMalloc(&dt,nt*8)); CK(hipMemcpy(dt,ht,nt*8,hipMemcpyHostToDevice));
  Entry es[] = {
    {"mont_exact/vgpr_tw", kern<0,false>}, {"mont_exact/sgpr_tw", kern<0,true>},
    {"shoup_approx/vgpr_tw", kern<1,false>}, {"shoup_approx/sgpr_tw", kern<1,true>},
    {"shoup_approx_mask/vgpr_tw", kern<2,false>}, {"shoup_approx_mask/sgpr_tw", kern<2,true>},
    {"shoup_exact/vgpr_tw", kern<3,false>}, {"shoup_exact/sgpr_tw", kern<3,true>},
  };
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 256;
  printf("%-28s %s\n", "variant", "blocks/CU -> Gbutterfly/s (cyc per wave-butterfly per SIMD @2.4GHz) [limb-NTT(2^16)/s equiv]");
  for(auto& e: es){
    printf("%-28s", e.name);
    for(int bpc: {1,2,4,8}){
      int blocks = cus*bpc;
      e.fn<<<blocks,256>>>(d,dt,m,2); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      e.fn<<<blocks,256>>>(d,dt,m,iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms,e0,e1));
      double bfl = (double)blocks*256*32*iters;
      double rate = bfl/(ms*1e-3);
      double cyc = (double)cus*4*2.4e9/(rate/64);
      printf(" | %d: %7.1f (%6.1f) [%.2fM]", bpc, rate*1e-9, cyc, rate/524288*1e-6);
    }
    printf("\n");
  }
  return 0;
}
