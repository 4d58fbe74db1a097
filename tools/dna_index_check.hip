// Device check of transform_base / dna_index against a host restatement for every character (tools/; run on the GPU box:
//   cd tools && hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -I../include -o /tmp/di dna_index_check.hip && /tmp/di).
// Written while chasing the k_prep_write discrepancy of round 2 (NOTES/rounds_1_to_4.md section 10): the function is identical on the device.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../dsrc_amd/csrc/k_common.h"
#include "../dsrc_amd/csrc/k_parse.h"
__global__ void k(const u8* in, const u8* qin, u32* out, u32* outq, u32 n, u32 lossy)
{
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	u32 sidx = 0, q = 0; bool keep = false;
	if (i < n) q = transform_base(in[i], qin[i], 33, lossy, &sidx, &keep);
	if (i < n) { out[i] = sidx | (keep ? 0x100u : 0u); outq[i] = q; }
}
static u32 lb(u32 q) { if (q < 2) return 0; if (q < 10) return 1; if (q < 20) return 2; if (q < 25) return 3; if (q < 30) return 4; if (q < 35) return 5; if (q < 40) return 6; if (q < 64) return 7; return 255; }
static u32 ref_q(u32 base, u32 qual, u32 qoff, u32 lossy, u32* sidx, bool* keep)
{
	const u32 s = dna_index_switch(base); *sidx = s; u32 q;
	if (!lossy) { q = (qual - qoff) & 255u; if (s > 3 && q < 7) { q = (q + 128u + ((s - 2u) << 3) - 16u) & 255u; *keep = false; } else *keep = true; }
	else { q = lb((qual - qoff) & 255u); if (s >= 4) { q = 0; *keep = false; } else { if (q == 0) q = 1; *keep = true; } }
	return q;
}
int main() {
	const u32 n = 1 << 16;
	u8* h = (u8*)malloc(n), *hq = (u8*)malloc(n);
	for (u32 i = 0; i < n; ++i) { h[i] = (i < 256) ? (u8)i : (u8)"ACGTNACGTRYKM.-acgt\n"[rand() % 20]; hq[i] = (u8)(33 + rand() % 42); }
	u8 *d, *dq; u32 *o, *oq; hipMalloc((void**)&d, n); hipMalloc((void**)&dq, n); hipMalloc((void**)&o, 4 * n); hipMalloc((void**)&oq, 4 * n);
	hipMemcpy(d, h, n, hipMemcpyHostToDevice); hipMemcpy(dq, hq, n, hipMemcpyHostToDevice);
	for (u32 lossy = 0; lossy < 2; ++lossy)
	{
		k<<<n / 256, 256>>>(d, dq, o, oq, n, lossy);
		u32* r = (u32*)malloc(4 * n), *rq = (u32*)malloc(4 * n); hipMemcpy(r, o, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(rq, oq, 4 * n, hipMemcpyDeviceToHost);
		int bad = 0;
		for (u32 i = 0; i < n; ++i)
		{
			u32 s; bool kp; const u32 q = ref_q(h[i], hq[i], 33, lossy, &s, &kp);
			if (r[i] != (s | (kp ? 0x100u : 0u)) || rq[i] != q) { if (bad < 8) printf("lossy=%u i=%u c=%u q=%u gpu=%x/%u want=%x/%u\n", lossy, i, h[i], hq[i], r[i], rq[i], s | (kp ? 0x100u : 0u), q); ++bad; }
		}
		printf("lossy=%u bad=%d\n", lossy, bad);
	}
	return 0;
}
