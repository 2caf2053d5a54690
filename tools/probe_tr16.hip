// probe: semantics of ds_read_b64_tr_b16 on gfx950 (which lane's address feeds which result element)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void probe(const int *byteoff, short *out) {
  __shared__ __attribute__((aligned(16))) short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x;
  s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)((char *)lds + byteoff[l]));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = t[j];
}
int main() {
  int h_off[64]; short h_out[256];
  int *d_off; short *d_out;
  hipMalloc(&d_off, sizeof(h_off)); hipMalloc(&d_out, sizeof(h_out));
  for (int pat = 0; pat < 3; ++pat) {
    for (int l = 0; l < 64; ++l) {
      if (pat == 0) h_off[l] = 8 * l;                       // lane-linear chunks
      else if (pat == 1) h_off[l] = 8 * ((l * 37 + 11) % 64) + 1024 * (l % 3);  // scrambled
      else h_off[l] = ((l >> 4) * 4 + ((l & 15) >> 2)) * 256 + (l & 3) * 8;      // rows of 256 B: row=(l>>4)*4+(l&15)/4, col chunk l&3
    }
    hipMemcpy(d_off, h_off, sizeof(h_off), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d_off, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("pattern %d\n", pat);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d off %5d ->", l, h_off[l]);
      for (int j = 0; j < 4; ++j) {
        int v = h_out[l * 4 + j];            // element index read
        // find which lane's chunk contains it
        int src = -1, pos = -1;
        for (int s = 0; s < 64; ++s) if (v * 2 >= h_off[s] && v * 2 < h_off[s] + 8) { src = s; pos = (v * 2 - h_off[s]) / 2; }
        printf("  e%5d(lane%2d.%d)", v, src, pos);
      }
      printf("\n");
    }
  }
  return 0;
}
