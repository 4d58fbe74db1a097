// Stand-alone reproducer for the round-2 k_prep_write discrepancy (task 5; NOTES/rounds_1_to_4.md section 10):
// the inner loop of k_prep_write (dsrc_amd/csrc/k_parse.h) in its two forms over synthetic records,
//   GOOD: transform_base<true>() evaluated by every lane, `in_r` applied afterwards (what ships),
//   BAD : transform_base<true>() called under `if (in_r)`                             (what miscompared on gfx950),
// each compared with a host restatement.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -o /tmp/pw tools/prep_write_repro.hip && /tmp/pw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../dsrc_amd/csrc/k_common.h"
#include "../dsrc_amd/csrc/k_parse.h"

template <bool BAD>
__global__ void __launch_bounds__(256) k(const u8* p, const u32* seq_off, const u32* qual_off, const u16* lens, const u32* q_off, const u32* d_off,
										 u32 n_recs, u8* q_stream, u8* qp_stream, u8* d_stream, u32 write_qp, u32 red, u32 qoff, u32 lossy)
{
	const u32 lane = lane_id();
	const u32 waves_total = gridDim.x * (blockDim.x >> 6);
	for (u32 r = blockIdx.x * (blockDim.x >> 6) + wave_id(); r < n_recs; r += waves_total)
	{
		const u32 rlen = lens[r], len = rlen + red, so = seq_off[r], qo = qual_off[r] - red;
		u8* qs = q_stream + q_off[r];
		u8* qps = qp_stream + q_off[r];
		u8* ds = d_stream + d_off[r];
		u32 run = 0;
		for (u32 j0 = 0; j0 < len; j0 += 64)
		{
			const u32 j = j0 + lane;
			const bool in_r = j < len;
			u32 sidx = 0, q = 0; bool keep = false;
			if (BAD) { if (in_r) q = transform_base<true>(p[so + j], p[qo + j], qoff, lossy, &sidx, &keep); }
			else
			{
				const u32 cb = in_r ? (u32)p[so + j] : (u32)'A', cq = in_r ? (u32)p[qo + j] : qoff + 40u;
				const u32 qq = transform_base<true>(cb, cq, qoff, lossy, &sidx, &keep);
				if (in_r) q = qq; else { keep = false; sidx = 0; }
			}
			const bool k2 = in_r && keep;
			const u64 km = __ballot(k2);
			if (in_r && j >= red) { qs[j - red] = (u8)q; if (write_qp) qps[j - red] = (u8)(((j - red) * 128u) / rlen); }
			const u32 at = run + (u32)__popcll(km & lanemask_lt());
			if (k2 && at >= red) ds[at - red] = (u8)sidx;
			run += (u32)__popcll(km);
		}
	}
}

static u32 host_q(u32 base, u32 qual, u32 qoff, u32* sidx, bool* keep)
{
	const u32 s = dna_index_switch_c(base); *sidx = s;
	u32 q = (qual - qoff) & 255u;
	if (s > 3 && q < 7) { q = (q + 128u + ((s - 2u) << 3) - 16u) & 255u; *keep = false; } else *keep = true;
	return q;
}

int main()
{
	srand(7);
	for (u32 variable = 0; variable < 2; ++variable)
	{
		const u32 n = 20000;
		std::vector<u8> text; std::vector<u32> so(n), qo(n), qoffs(n), doffs(n); std::vector<u16> lens(n);
		std::vector<u8> want_q, want_d;
		for (u32 r = 0; r < n; ++r)
		{
			const u32 len = variable ? 1 + rand() % 300 : 150;
			lens[r] = (u16)len; so[r] = (u32)text.size();
			std::vector<u8> b(len), q(len);
			for (u32 i = 0; i < len; ++i) { const bool isn = rand() % 500 == 0; b[i] = isn ? 'N' : "ACGT"[rand() % 4]; q[i] = (u8)(33 + (isn ? 2 : 2 + rand() % 39)); }
			text.insert(text.end(), b.begin(), b.end()); text.push_back('\n');
			qo[r] = (u32)text.size(); text.insert(text.end(), q.begin(), q.end()); text.push_back('\n');
			qoffs[r] = (u32)want_q.size(); doffs[r] = (u32)want_d.size();
			for (u32 i = 0; i < len; ++i) { u32 s; bool kp; want_q.push_back((u8)host_q(b[i], q[i], 33, &s, &kp)); if (kp) want_d.push_back((u8)s); }
		}
		u8 *d_p, *d_q, *d_qp, *d_d; u32 *d_so, *d_qo, *d_qoff, *d_doff; u16* d_len;
		hipMalloc((void**)&d_p, text.size() + 64); hipMalloc((void**)&d_q, want_q.size() + 64); hipMalloc((void**)&d_qp, want_q.size() + 64); hipMalloc((void**)&d_d, want_d.size() + 64);
		hipMalloc((void**)&d_so, 4 * n); hipMalloc((void**)&d_qo, 4 * n); hipMalloc((void**)&d_qoff, 4 * n); hipMalloc((void**)&d_doff, 4 * n); hipMalloc((void**)&d_len, 2 * n);
		hipMemcpy(d_p, text.data(), text.size(), hipMemcpyHostToDevice);
		hipMemcpy(d_so, so.data(), 4 * n, hipMemcpyHostToDevice); hipMemcpy(d_qo, qo.data(), 4 * n, hipMemcpyHostToDevice);
		hipMemcpy(d_qoff, qoffs.data(), 4 * n, hipMemcpyHostToDevice); hipMemcpy(d_doff, doffs.data(), 4 * n, hipMemcpyHostToDevice); hipMemcpy(d_len, lens.data(), 2 * n, hipMemcpyHostToDevice);
		for (u32 bad = 0; bad < 2; ++bad)
		{
			hipMemset(d_q, 0xEE, want_q.size()); hipMemset(d_d, 0xEE, want_d.size());
			if (bad) hipLaunchKernelGGL(k<true>, dim3(64), dim3(256), 0, 0, d_p, d_so, d_qo, d_len, d_qoff, d_doff, n, d_q, d_qp, d_d, variable, 0u, 33u, 0u);
			else hipLaunchKernelGGL(k<false>, dim3(64), dim3(256), 0, 0, d_p, d_so, d_qo, d_len, d_qoff, d_doff, n, d_q, d_qp, d_d, variable, 0u, 33u, 0u);
			std::vector<u8> gq(want_q.size()), gd(want_d.size());
			hipMemcpy(gq.data(), d_q, gq.size(), hipMemcpyDeviceToHost); hipMemcpy(gd.data(), d_d, gd.size(), hipMemcpyDeviceToHost);
			size_t bq = 0, bd = 0, first = (size_t)-1;
			for (size_t i = 0; i < gq.size(); ++i) if (gq[i] != want_q[i]) { if (first == (size_t)-1) first = i; ++bq; }
			for (size_t i = 0; i < gd.size(); ++i) if (gd[i] != want_d[i]) ++bd;
			printf("%s lengths, %s form: quality stream %zu wrong of %zu, base stream %zu wrong of %zu", variable ? "variable" : "constant", bad ? "BAD " : "GOOD", bq, gq.size(), bd, gd.size());
			if (bq) printf("  (first at %zu: got %02x want %02x)", first, gq[first], want_q[first]);
			printf("\n");
		}
	}
	return 0;
}
