/*
 * wtz_testdp.h — wtz_test_dp(): the TEST-ONLY entry of the C ABI (include/wtzmo_hip.h).  It runs single banded-DP problems through a
 * chosen device form of K-sw1 / K-sw2 / K-sw3 so that every form can be compared, function by function, with vectors dumped from the
 * reference's own kswx_extend_align_core (kswx.h:234), kswx_extend_align_shift_core (kswx.h:101) and ksw_global2 (ksw.c:503).
 * The forms are selected by the very functions the product kernels call (wtz_fixed_problem_wave, wtz_gap_problem_wave, run_extjobs and
 * the extension-job kernels); this file only wraps problems into their inputs and collects the results.  Included by wtz_lib.cpp.
 */
#ifndef WTZ_TESTDP_H
#define WTZ_TESTDP_H

struct K_test_fixed;
struct K_test_global;
struct K_test_global_wide;
struct K_test_lane;

typedef struct { wtz_seq_packed q, t; int32_t qlen, tlen, init_score, W; } wtz_dpprob_dev_t;
typedef struct { wtz_aln_t x; uint32_t *cigar; uint32_t cigar_len; int32_t form, bad; unsigned long long cells; } wtz_dpres_dev_t;

#ifndef WTZ_EMUL
static __device__ void wtz_testdp_store(wtz_dpres_dev_t *res, const wtz_aln_t &x, bool from_runs, const uint32_t *runs, uint32_t n_runs, const wtz_cigar_t &tmp,
		wtz_pool_t *pool, int form, int bad, unsigned long long cells){
	/* lane 0 only */
	wtz_dpres_dev_t r; memset(&r, 0, sizeof r);
	r.x = x; r.form = form; r.bad = bad; r.cells = cells;
	if(from_runs){
		wtz_cigar_t cg; cg.init(pool, n_runs ? n_runs : 1);
		for(uint32_t k = n_runs; k-- > 0;) cg.push(runs[k]);         /* runs are in traceback order */
		r.cigar = cg.a; r.cigar_len = cg.n; if(cg.bad) r.bad = 1;
	} else { r.cigar = tmp.a; r.cigar_len = tmp.n; if(tmp.bad) r.bad = 1; }
	*res = r;
}

static __device__ void wtz_task_test_fixed(uint32_t t, const wtz_dpprob_dev_t *pr, const wtz_params_t *P, wtz_pool_t *pool, int force, wtz_dpres_dev_t *res){
	const wtz_dpprob_dev_t p = pr[t];
	int32_t *lds = wtz_wave_scratch();
	wtz_wave_lds_t L; L.tb = (uint64_t*)lds; L.Hs = lds + 256; L.Es = lds + 768; L.PM = 511; L.tw = 128; L.qw = (uint32_t*)(lds + WTZ_WINALIGN_LDS_BYTES / 4);        /* the slice of wtz_align_window_wave */
	uint8_t *ztr = (uint8_t*)(lds + 256); const int32_t ztr_bytes = WTZ_WINALIGN_LDS_BYTES - 1024;
	wtz_swmem_t mem; wtz_swmem_init(mem, pool);
	wtz_cigar_t tmp; tmp.init(pool, WTZ_LANE == 0 ? 64 : 0);
	unsigned long long cells = 0;
	wtz_fixres_t R;
	wtz_fixed_problem_wave<true, true>(p.qlen, p.q, p.tlen, p.t, p.init_score, P, L, ztr, ztr_bytes, tmp, mem, pool, &cells, force, R);
	if(WTZ_LANE != 0) return;
	if(R.form < 0){ wtz_dpres_dev_t r; memset(&r, 0, sizeof r); r.form = 0; res[t] = r; return; }
	/* an empty problem is answered in place by every form (score = init, empty CIGAR): report it under the requested name */
	wtz_testdp_store(&res[t], R.y, R.lds_runs, R.runs, R.n_runs, tmp, pool, R.form ? R.form : (force ? force : 1), !R.ok, cells);
}

/* form 64: the lane-per-problem K-sw1 (wtz_sw_lane.h) in ABSOLUTE mode - the reference's function for a known init_score; lane 0 of the wave
 * carries the problem, the other lanes are idle (the product packs 64 problems of one shape class into a wave and runs the relative mode) */
static __device__ void wtz_task_test_lane(uint32_t t, const wtz_dpprob_dev_t *pr, const wtz_params_t *P, wtz_pool_t *pool, wtz_dpres_dev_t *res){
	const wtz_dpprob_dev_t p = pr[t];
	const int32_t M = P->M, X = P->X, I = P->O, D = P->O, E = P->E, T = P->T;
	const int32_t init = p.init_score < 0 ? 0 : p.init_score;
	wtz_dpres_dev_t r; memset(&r, 0, sizeof r);
	if(p.qlen <= 0 || p.tlen <= 0){ r.x.score = init; r.form = 64; if(WTZ_LANE == 0) res[t] = r; return; }
	int32_t W = P->w, ql, tl, n_col;
	wtz_ext_geometry(p.qlen, p.tlen, init, W, M, I, D, E, T, ql, tl, n_col);
	if(n_col > WTZ_LN_MAXCOLS || ql > WTZ_LN_MAXROWS || ql + tl > WTZ_LN_MAXSPAN || init > (1 << 20)){ if(WTZ_LANE == 0) res[t] = r; return; }      /* outside the envelope: declined */
	const uint32_t RS = wtz_lane_rs(n_col);
	unsigned long long pa = 0;
	if(WTZ_LANE == 0) pa = (unsigned long long)(uintptr_t)wtz_pool_alloc(pool, (size_t)ql * RS * 4 * WTZ_NLANES + (size_t)(ql + tl + 4) * 4 + 32);      /* the wave's lane-interleaved trace block (wtz_ltr_at), although only lane 0 has a problem here */
	pa = __shfl(pa, 0, 64);
	if(pa == 0){ r.bad = 1; r.form = 64; if(WTZ_LANE == 0) res[t] = r; return; }
	uint32_t *tr = (uint32_t*)(uintptr_t)pa, *runs = tr + (size_t)ql * RS * WTZ_NLANES + 4;
	const bool live = WTZ_LANE == 0;
	wtz_lres_t R; memset(&R, 0, sizeof R);
	if(n_col <= 16)      wtz_lane_fixed<16, true>(live, p.qlen, p.q, p.tlen, p.t, init, W, ql, tl, M, X, I, D, E, T, tr, R);
	else if(n_col <= 32) wtz_lane_fixed<32, true>(live, p.qlen, p.q, p.tlen, p.t, init, W, ql, tl, M, X, I, D, E, T, tr, R);
	else if(n_col <= 64) wtz_lane_fixed<64, true>(live, p.qlen, p.q, p.tlen, p.t, init, W, ql, tl, M, X, I, D, E, T, tr, R);
	else                 wtz_lane_fixed<104, true>(live, p.qlen, p.q, p.tlen, p.t, init, W, ql, tl, M, X, I, D, E, T, tr, R);
	if(!live) return;
	wtz_lane_traceback<false>(true, R.qe - 1, R.te - 1, W, RS, p.q, p.qlen, p.t, p.tlen, tr, runs, R);
	wtz_aln_t x; memset(&x, 0, sizeof x);
	x.score = R.score; x.qe = R.qe; x.te = R.te; x.mat = R.mat; x.mis = R.mis; x.ins = R.ins; x.del = R.del; x.aln = R.mat + R.mis + R.ins + R.del;
	wtz_cigar_t none; none.a = NULL; none.n = none.cap = 0; none.pool = pool; none.bad = 0;
	wtz_testdp_store(&res[t], x, true, runs, R.n_runs, none, pool, 64, 0, R.cells);
}

/* form 64 of WTZ_DP_GLOBAL: the lane-per-gap K-sw2 (wtz_lane_global) at the band width the problem names */
static __device__ void wtz_task_test_lane_global(uint32_t t, const wtz_dpprob_dev_t *pr, const wtz_params_t *P, wtz_pool_t *pool, wtz_dpres_dev_t *res){
	const wtz_dpprob_dev_t p = pr[t];
	const int32_t M = P->M, X = P->X, I = P->O, D = P->O, E = P->E;
	wtz_dpres_dev_t r; memset(&r, 0, sizeof r);
	const int32_t w = p.W, n_col = p.qlen < 2 * w + 1 ? p.qlen : 2 * w + 1;
	if(p.qlen <= 0 || p.tlen <= 0 || WTZ_ABSDIFF(p.qlen, p.tlen) > w || n_col > WTZ_LN_MAXCOLS || p.tlen > WTZ_LG_MAXROWS){ if(WTZ_LANE == 0) res[t] = r; return; }     /* declined */
	const uint32_t RS = wtz_lane_rs(n_col);
	unsigned long long pa = 0;
	if(WTZ_LANE == 0) pa = (unsigned long long)(uintptr_t)wtz_pool_alloc(pool, (size_t)p.tlen * RS * 4 * WTZ_NLANES + (size_t)(p.qlen + p.tlen + 4) * 4 + 32);
	pa = __shfl(pa, 0, 64);
	if(pa == 0){ r.bad = 1; r.form = 64; if(WTZ_LANE == 0) res[t] = r; return; }
	uint32_t *tr = (uint32_t*)(uintptr_t)pa, *runs = tr + (size_t)p.tlen * RS * WTZ_NLANES + 4;
	const bool live = WTZ_LANE == 0;
	wtz_lres_t R; memset(&R, 0, sizeof R);
	if(n_col <= 16)      wtz_lane_global<16>(live, p.qlen, p.q, p.tlen, p.t, w, M, X, -I, -E, -D, -E, tr, R);
	else if(n_col <= 32) wtz_lane_global<32>(live, p.qlen, p.q, p.tlen, p.t, w, M, X, -I, -E, -D, -E, tr, R);
	else if(n_col <= 64) wtz_lane_global<64>(live, p.qlen, p.q, p.tlen, p.t, w, M, X, -I, -E, -D, -E, tr, R);
	else                 wtz_lane_global<104>(live, p.qlen, p.q, p.tlen, p.t, w, M, X, -I, -E, -D, -E, tr, R);
	if(!live) return;
	{ const int32_t r0 = p.tlen - 1, c0 = (r0 + w + 1 < p.qlen ? r0 + w + 1 : p.qlen) - 1; wtz_lane_traceback<true>(true, r0, c0, w, RS, p.t, p.tlen, p.q, p.qlen, tr, runs, R); }
	wtz_aln_t x; memset(&x, 0, sizeof x);
	x.score = R.score; x.mat = R.mat; x.mis = R.mis; x.ins = R.ins; x.del = R.del; x.aln = R.mat + R.mis + R.ins + R.del;
	wtz_cigar_t none; none.a = NULL; none.n = none.cap = 0; none.pool = pool; none.bad = 0;
	wtz_testdp_store(&res[t], x, true, runs, R.n_runs, none, pool, 64, 0, R.cells);
}

static __device__ void wtz_task_test_global(uint32_t t, const wtz_dpprob_dev_t *pr, const wtz_params_t *P, wtz_pool_t *pool, int force, uint32_t wide_lds, wtz_dpres_dev_t *res){
#if defined(__HIP_DEVICE_COMPILE__)
	const wtz_dpprob_dev_t p = pr[t];
	int32_t *lds = wtz_wave_scratch();
	wtz_trace_t tr; tr.chunk = NULL; tr.zb = NULL; tr.n_chunk = 0; tr.zrow = 0; tr.cap_rows = 0;
	wtz_swmem_t mem; wtz_swmem_init(mem, pool);
	wtz_cigar_t tmp; tmp.init(pool, WTZ_LANE == 0 ? 32 : 0);
	wtz_gapdp_t G;
	wtz_gap_problem_wave<true>(p.qlen, p.q, p.tlen, p.t, P, p.W, lds, wide_lds, false, pool, tmp, tr, mem, force == 33 ? WTZ_FORM_RING : force, G);
	if(WTZ_LANE != 0) return;
	if(G.form < 0){ wtz_dpres_dev_t r; memset(&r, 0, sizeof r); r.form = 0; res[t] = r; return; }
	wtz_aln_t x; memset(&x, 0, sizeof x); x.score = G.score;
	if(G.from_reg){
		x.mat = G.r_mat; x.mis = G.r_mis;
		for(uint32_t k = 0; k < G.n_runs; k++){ const uint32_t r = G.runs[k], op = r & 0xFu; const int32_t len = (int32_t)(r >> 4); x.aln += len; if(op == 1) x.ins += len; else if(op == 2) x.del += len; }
	} else {        /* the fold of wtz_task_gap over a CIGAR that came without counts */
		int32_t x1 = 0, x2 = 0;
		for(uint32_t idx = 0; idx < tmp.n; idx++){
			const int32_t op = (int32_t)(tmp.a[idx] & 0xF), len = (int32_t)(tmp.a[idx] >> 4);
			x.aln += len;
			if(op == 0){ for(int32_t j = 0; j < len; j++){ if(p.q.at(x1 + j) == p.t.at(x2 + j)) x.mat++; else x.mis++; } x1 += len; x2 += len; }
			else if(op == 1){ x1 += len; x.ins += len; }
			else if(op == 2){ x2 += len; x.del += len; }
		}
	}
	wtz_testdp_store(&res[t], x, G.from_reg, G.runs, G.n_runs, tmp, pool, (G.form == WTZ_FORM_RING && force == 33) ? 33 : G.form, G.bad, 0);
#endif
}
#endif

extern "C" int wtz_test_dp(wtz_ctx_t *c, int32_t kind, int32_t form, const wtz_dp_problem_t *pr, uint32_t n, wtz_dp_result_t *out, uint32_t *cigar, uint64_t cigar_cap){
	if(!c || !c->bits) return wtz_fail(WTZ_E_ARG, "reads not uploaded");
	if(n == 0) return WTZ_OK;
	if(!pr || !out || (!cigar && cigar_cap)) return wtz_fail(WTZ_E_ARG, "null argument");
#ifdef WTZ_EMUL
	(void)kind; (void)form;
	return wtz_fail(WTZ_E_STATE, "wtz_test_dp drives the DEVICE forms of the banded DPs: it needs the HIP build");
#else
	CTX_ENTER(c);
	CHK(pool_reset(c));
	std::vector<wtz_dpprob_dev_t> hp(n);
	std::vector<uint64_t> h_off(c->n_reads);
	CHK(dev_d2h(h_off.data(), c->rdoff, (size_t)c->n_reads * 8));
	for(uint32_t i = 0; i < n; i++){
		const wtz_dp_problem_t &p = pr[i];
		if(p.q_read >= c->n_reads || p.t_read >= c->n_reads) return wtz_fail(WTZ_E_ARG, "problem %u: read id out of range", i);
		if((p.q_strand != 1 && p.q_strand != -1) || (p.t_strand != 1 && p.t_strand != -1)) return wtz_fail(WTZ_E_ARG, "problem %u: strand must be +1 or -1", i);
		wtz_readview vq, vt;
		vq.bits = c->bits; vq.off = h_off[p.q_read]; vq.len = c->h_rdlen[p.q_read]; vq.rev = p.q_rev ? 1u : 0u;
		vt.bits = c->bits; vt.off = h_off[p.t_read]; vt.len = c->h_rdlen[p.t_read]; vt.rev = p.t_rev ? 1u : 0u;
		const int64_t qlast = (int64_t)p.q_from + (int64_t)p.q_strand * (p.q_len > 0 ? p.q_len - 1 : 0), tlast = (int64_t)p.t_from + (int64_t)p.t_strand * (p.t_len > 0 ? p.t_len - 1 : 0);
		if(p.q_len < 0 || p.t_len < 0 || (p.q_len > 0 && (p.q_from < 0 || p.q_from >= (int64_t)vq.len || qlast < 0 || qlast >= (int64_t)vq.len))
				|| (p.t_len > 0 && (p.t_from < 0 || p.t_from >= (int64_t)vt.len || tlast < 0 || tlast >= (int64_t)vt.len)))
			return wtz_fail(WTZ_E_ARG, "problem %u: region outside its read", i);
		wtz_dpprob_dev_t d; d.q = vq.sub(p.q_from, p.q_strand); d.t = vt.sub(p.t_from, p.t_strand); d.qlen = p.q_len; d.tlen = p.t_len; d.init_score = p.init_score; d.W = p.W;
		hp[i] = d;
	}
	const wtz_env_t V = ctx_env(c);
	std::vector<wtz_dpres_dev_t> hr(n);
	if(kind == WTZ_DP_SHIFT){
		std::vector<wtz_extjob_t> jobs(n);
		for(uint32_t i = 0; i < n; i++){ wtz_extjob_t j; memset(&j, 0, sizeof j); j.q = hp[i].q; j.t = hp[i].t; j.qlen = hp[i].qlen; j.tlen = hp[i].tlen; j.init_score = hp[i].init_score; j.W = hp[i].W; j.item = i; j.valid = 1; jobs[i] = j; }
		wtz_extjob_t *d_jobs = NULL; CHK(dev_alloc((void**)&d_jobs, (size_t)n * sizeof(wtz_extjob_t))); CHK(dev_h2d(d_jobs, jobs.data(), (size_t)n * sizeof(wtz_extjob_t)));
		wtz_timer tform; tform.start();      /* the forced forms report their launch time through counters.ms_ext too (tools/ubench/ksw3_bench.py) */
		if(form == 0){ CHK(run_extjobs(c, V, d_jobs, n)); }
		else if(form == 1){ hipLaunchKernelGGL((wtz_kernel_extjobs_reg<1032>), dim3(n), dim3(64), 0, g_stream, d_jobs, (const uint32_t*)NULL, n, V.P, V.pool, V.pool + 1); HIPCHK(hipGetLastError()); }
		else if(form == 3){ hipLaunchKernelGGL((wtz_kernel_extjobs<2048, 1032>), dim3(n), dim3(64), 0, g_stream, d_jobs, (const uint32_t*)NULL, n, V.P, V.pool, V.pool + 1); HIPCHK(hipGetLastError()); }
		else if(form == 4){ CHK(wtz_launch_wave<K_extjob_scalar>(0, n, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_extjob_scalar((uint32_t)t, V, d_jobs); })); }
		else if(form == 5){ hipLaunchKernelGGL((wtz_kernel_extjobs_fr<1032>), dim3(n), dim3(64), 0, g_stream, d_jobs, (const uint32_t*)NULL, n, V.P, V.pool, V.pool + 1); HIPCHK(hipGetLastError()); }
		else if(form == 6){ hipLaunchKernelGGL((wtz_kernel_extjobs_frmw<1032>), dim3(n), dim3(256), 0, g_stream, d_jobs, (const uint32_t*)NULL, n, V.P, V.pool, V.pool + 1); HIPCHK(hipGetLastError()); }      /* frame form on four wavefronts (round 6) */
		else if(form == 7){ hipLaunchKernelGGL((wtz_kernel_extjobs_pk<1032>), dim3(n), dim3(64), 0, g_stream, d_jobs, (const uint32_t*)NULL, n, V.P, V.pool, V.pool + 1); HIPCHK(hipGetLastError()); }      /* frame form, two 16-bit cells per register (round 6) */
		else return wtz_fail(WTZ_E_ARG, "WTZ_DP_SHIFT: unknown form %d", form);
		CHK(dev_sync());
		if(form != 0){ c->cnt.ms_ext += tform.stop(); c->cnt.n_extjobs += n; }
		CHK(tpool_check(c, "wtz_test_dp"));
		CHK(dev_d2h(jobs.data(), d_jobs, (size_t)n * sizeof(wtz_extjob_t)));
		for(uint32_t i = 0; i < n; i++){
			wtz_dpres_dev_t r; memset(&r, 0, sizeof r);
			const int done = form == 4 ? 4 : (int)jobs[i].done;
			const bool empty = jobs[i].qlen <= 0 || jobs[i].tlen <= 0;
			/* the two register kernels leave empty problems (and what is outside their envelope) to the general kernel */
			if(done || (form == 0) || (form == 3 && empty)){ r.x = jobs[i].x; r.cigar = jobs[i].cigar; r.cigar_len = jobs[i].cigar_len; r.form = done ? done : 3; r.bad = jobs[i].bad; r.cells = jobs[i].cells; }
			hr[i] = r;
		}
	} else if(kind == WTZ_DP_FIXED || kind == WTZ_DP_GLOBAL){
		wtz_dpprob_dev_t *d_pr = NULL; wtz_dpres_dev_t *d_res = NULL;
		CHK(dev_alloc((void**)&d_pr, (size_t)n * sizeof(wtz_dpprob_dev_t))); CHK(dev_h2d(d_pr, hp.data(), (size_t)n * sizeof(wtz_dpprob_dev_t)));
		CHK(dev_alloc((void**)&d_res, (size_t)n * sizeof(wtz_dpres_dev_t))); CHK(dev_set(d_res, 0, (size_t)n * sizeof(wtz_dpres_dev_t)));
		const wtz_params_t *dP = c->dP; wtz_pool_t *pool = c->dpool;
		if(kind == WTZ_DP_FIXED && form == 64){
			CHK(wtz_launch_coop<K_test_lane>(0, n, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_test_lane((uint32_t)t, d_pr, dP, pool, d_res); }, 0));
		} else if(kind == WTZ_DP_FIXED){
			CHK(wtz_launch_coop<K_test_fixed>(0, n, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_test_fixed((uint32_t)t, d_pr, dP, pool, form, d_res); }, WTZ_WINALIGN_LDS_BYTES + WTZ_WINALIGN_QW_BYTES));
		} else if(form == 64){
			CHK(wtz_launch_coop<K_test_lane>(0, n, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_test_lane_global((uint32_t)t, d_pr, dP, pool, d_res); }, 0));
		} else if(form == 33){
			CHK(wtz_launch_coop<K_test_global_wide>(0, n, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_test_global((uint32_t)t, d_pr, dP, pool, form, (uint32_t)WTZ_GAP_WIDE_LDS_BYTES, d_res); }, WTZ_GAP_WIDE_LDS_BYTES));
		} else {
			CHK(wtz_launch_coop<K_test_global>(0, n, [=] WTZ_LAMBDA (uint64_t t){ wtz_task_test_global((uint32_t)t, d_pr, dP, pool, form, 0u, d_res); }, WTZ_GAP_LDS_BYTES));
		}
		CHK(dev_sync());
		CHK(dev_d2h(hr.data(), d_res, (size_t)n * sizeof(wtz_dpres_dev_t)));
	} else return wtz_fail(WTZ_E_ARG, "unknown DP kind %d", kind);
	CHK(pool_check(c, "wtz_test_dp"));
	uint64_t off = 0;
	for(uint32_t i = 0; i < n; i++){
		const wtz_dpres_dev_t &r = hr[i];
		if(r.bad) return wtz_fail(WTZ_E_POOL, "wtz_test_dp: problem %u ran out of scratch", i);
		wtz_dp_result_t o; memset(&o, 0, sizeof o);
		o.form_used = (uint32_t)r.form;
		if(r.form){
			o.score = r.x.score; o.tb = r.x.tb; o.te = r.x.te; o.qb = r.x.qb; o.qe = r.x.qe; o.aln = r.x.aln; o.mat = r.x.mat; o.mis = r.x.mis; o.ins = r.x.ins; o.del = r.x.del;
			o.cigar_len = r.cigar_len; o.cigar_off = off; o.cells = r.cells;
			if(off + r.cigar_len > cigar_cap) return wtz_fail(WTZ_E_ARG, "wtz_test_dp: CIGAR buffer too small");
			if(r.cigar_len) CHK(dev_d2h(cigar + off, r.cigar, (size_t)r.cigar_len * 4));
			off += r.cigar_len;
		}
		out[i] = o;
	}
	return WTZ_OK;
#endif
}
#endif
