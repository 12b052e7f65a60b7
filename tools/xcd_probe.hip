// Dev tool: which XCD does each workgroup of a 3-D grid land on?  (HW_REG_XCC_ID, see MI355X_MICROARCH.md)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    out[lin] = (int)(xcc & 0xf);
  }
}
int main() {
  dim3 grid(5, 8, 3);
  const int n = grid.x * grid.y * grid.z;
  int* d; int h[512];
  hipMalloc(&d, n * 4);
  k<<<grid, 64>>>(d);
  hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
  int ok = 0;
  for (int i = 0; i < n; ++i) ok += (h[i] == h[i % 8]);
  printf("grid (5,8,3): %d of %d blocks satisfy xcc(lin) == xcc(lin %% 8) with lin = x + gx*(y + gy*z)\n", ok, n);
  for (int i = 0; i < 24; ++i) printf("%d ", h[i]);
  printf("\n");
  return 0;
}
