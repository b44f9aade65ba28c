// micro_register.hip -- does hipHostRegister / hipHostUnregister of ordinary (malloc'd) memory followed by free + re-use of the address range by
// a pageable hipMemcpy upset the runtime?  (suspected cause of a silent abort inside rh_ring_create_auto right after a test that registered a
// numpy array.)  Bounded: 300 rounds, sizes from 6 KiB (unaligned heap) to 4 MiB (mmap'd).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
int main() {
  char* d; hipMalloc((void**)&d, 64 << 20);
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  const size_t sizes[] = {6144, 160 << 10, 256 << 10, 1 << 20, 4 << 20};
  for (int round = 0; round < 300; ++round) {
    const size_t n = sizes[round % 5];
    char* p = (char*)malloc(n);
    memset(p, round, n);
    hipError_t e = hipHostRegister(p, n, hipHostRegisterDefault);
    if (e != hipSuccess) { printf("round %d: register(%zu) -> %s\n", round, n, hipGetErrorString(e)); (void)hipGetLastError(); free(p); continue; }
    hipMemcpyAsync(d, p, n, hipMemcpyHostToDevice, st);
    hipMemcpyAsync(p, d, n, hipMemcpyDeviceToHost, st);
    hipStreamSynchronize(st);
    e = hipHostUnregister(p);
    if (e != hipSuccess) printf("round %d: unregister -> %s\n", round, hipGetErrorString(e));
    free(p);
    // what a ring construction does next: big pageable vectors uploaded with synchronous copies, likely in the address range just freed
    std::vector<char> v((size_t)4 << 20, (char)round);
    char* d2; hipMalloc((void**)&d2, v.size());
    hipMemcpy(d2, v.data(), v.size(), hipMemcpyHostToDevice);
    hipFree(d2);
    if (round % 50 == 49) { printf("round %d ok\n", round); fflush(stdout); }
  }
  printf("done: no abort\n");
  return 0;
}
