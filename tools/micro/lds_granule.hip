// How many 256-thread blocks with N bytes of static LDS does the runtime place on one CU of gfx950 (160 KB of LDS)?
//   hipcc --offload-arch=gfx950 -O2 tools/micro/lds_granule.hip -o build/var/lds_granule && build/var/lds_granule
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> __global__ void __launch_bounds__(256) k(double *out)
{
    __shared__ char s[N];
    s[threadIdx.x] = (char)threadIdx.x;
    __syncthreads();
    out[threadIdx.x] = s[(threadIdx.x * 7) % N];
}
template <int N> void probe()
{
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k<N>, 256, 0);
    printf("static LDS %6d B: %d blocks per CU (%d B in use)\n", N, nb, nb * N);
}
int main()
{
    probe<31744>(); probe<32000>(); probe<32256>(); probe<32488>(); probe<32768>(); probe<32769>(); probe<33280>();
    probe<40448>(); probe<40680>(); probe<40960>(); probe<40961>(); probe<27136>(); probe<27307>(); probe<27392>();
    return 0;
}
