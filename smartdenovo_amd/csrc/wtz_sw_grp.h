/*
 * wtz_sw_grp.h — A9 (fast_seeds_align_hzmo, hzm_aln.h:1247-1302) with FOUR chain windows per wavefront.
 *
 * A window is a chain of ~13 small K-sw1 problems (kswx_extend_align_core, kswx.h:234-335: tens of rows, a band of <= 2w+1 = 101
 * columns) tied together by the running score, each followed by a strictly sequential traceback, CIGAR fold and z-mer run
 * alignment.  One 64-lane wave per window (wtz_align_window_wave) spends half of its instructions in those single-lane sections
 * and uses at most 101 of its 128 column slots in the rows.  Here a wave is cut into four GROUPS of 16 lanes - exactly the DPP
 * "row" of the hardware, so every cross-lane step of a group (neighbour hand-over, max-plus scan of F, arg-max reduction) is a
 * row_shl / row_shr / row_ror DPP move that cannot see the other groups - and each group runs its own window: its own anchor
 * loop, rows, traceback and fold, in ordinary SIMT divergence.  Lane l of a group owns the C = 8 band-relative columns
 * l*C .. l*C+C-1 (128 columns >= 2w+1 for w <= 63).
 *
 * Per group, LDS holds the window's two sequence spans (staged once: no global load per problem), a 2 KB stage for the traceback
 * and the run list; the 4-bit trace (two rows per byte, 8 bytes per lane and row pair = one 64-bit store) goes to the pool and
 * comes back through the stage in blocks of 32 rows.  Whatever is outside this envelope (wider band, longer problem, longer span,
 * scores beyond the packed arg-max key) makes the group hand its window to the one-window-per-wave kernel (defer list): results
 * are those of wtz_extend_fixed_reg / kswx_extend_align_core either way.
 */
#ifndef WTZ_SW_GRP_H
#define WTZ_SW_GRP_H

#ifdef __HIPCC__

#define WTZ_GRP_LANES 16
#define WTZ_GRP_C 8                                         /* band columns per lane */
#define WTZ_GRP_MAXSPAN 1520                                /* bases of each read a window may span */
#define WTZ_GRP_SEQ_U32 (WTZ_GRP_MAXSPAN / 16 + 3)          /* 16 bases per u32 (+ the words a 16-base fetch at the end touches) */
#define WTZ_GRP_STAGE_BYTES 2048
#define WTZ_GRP_RUNS 508
#define WTZ_GRP_ZROW (WTZ_GRP_LANES * 8)                    /* trace bytes per row pair */
#define WTZ_GRP_LDS_BYTES (2 * WTZ_GRP_SEQ_U32 * 4 + WTZ_GRP_STAGE_BYTES + WTZ_GRP_RUNS * 4)
#define WTZ_WINALIGN4_LDS_BYTES (4 * WTZ_GRP_LDS_BYTES)

/* ---- group (DPP row) primitives: lanes of one row of 16 only ---- */
template<int CTRL> WTZ_D int32_t wtz_row_dpp(int32_t old, int32_t src){ return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xF, 0xF, false); }
WTZ_D int32_t wtz_grp_from_next(int32_t old, int32_t src){ return wtz_row_dpp<0x101>(old, src); }      /* row_shl:1 - lane l gets lane l+1, the last lane keeps `old` */
WTZ_D int32_t wtz_grp_from_prev(int32_t old, int32_t src){ return wtz_row_dpp<0x111>(old, src); }      /* row_shr:1 - lane l gets lane l-1, the first lane keeps `old` */
WTZ_D int32_t wtz_grp_max_scan_excl(int32_t v, int32_t ident){
	int32_t x = v, t;
	t = wtz_row_dpp<0x111>(ident, x); x = x > t ? x : t;
	t = wtz_row_dpp<0x112>(ident, x); x = x > t ? x : t;
	t = wtz_row_dpp<0x114>(ident, x); x = x > t ? x : t;
	t = wtz_row_dpp<0x118>(ident, x); x = x > t ? x : t;
	return wtz_row_dpp<0x111>(ident, x);
}
WTZ_D int32_t wtz_grp_max_all(int32_t v){                  /* every lane of the row gets the row maximum (rotations) */
	int32_t x = v, t;
	t = wtz_row_dpp<0x128>(x, x); x = x > t ? x : t;
	t = wtz_row_dpp<0x124>(x, x); x = x > t ? x : t;
	t = wtz_row_dpp<0x122>(x, x); x = x > t ? x : t;
	t = wtz_row_dpp<0x121>(x, x); x = x > t ? x : t;
	return x;
}
WTZ_D int32_t wtz_grp_min_all(int32_t v){ return -wtz_grp_max_all(-v); }
WTZ_D int32_t wtz_grp_lane(int32_t v, int src){ return __shfl(v, src, WTZ_GRP_LANES); }                /* value of lane `src` of the own group */
WTZ_D unsigned long long wtz_grp_lane64(unsigned long long v, int src){
	const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, WTZ_GRP_LANES), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, WTZ_GRP_LANES);
	return ((unsigned long long)hi << 32) | lo;
}

/* 16 bases starting at base p of a sequence staged as u32 words (base k of word w at bits 2k) */
WTZ_D uint32_t wtz_lds16(const uint32_t *s32, int32_t p){
	const uint32_t lo = s32[p >> 4], hi = s32[(p >> 4) + 1];
	return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)(p & 15) * 2u);
}
struct wtz_seq_lds { const uint32_t *s32; int32_t off; WTZ_D uint32_t at(int32_t i) const { const int32_t p = off + i; return (s32[p >> 4] >> ((p & 15) * 2)) & 3u; } };

/*
 * K-sw1 on one group.  All 16 lanes call with identical arguments; the result is complete on lane 0 of the group (score, qe, te
 * are identical on every lane).  q32 / t32: the staged sequences, qo / to: base offset of the problem's first base in them.
 * ztr: (ql+1)/2 row pairs of WTZ_GRP_ZROW bytes in the pool (lane l owns bytes l*8 .. l*8+C-1).  runs: traceback-order run list.
 */
template<int C>
WTZ_D wtz_aln_t wtz_extend_fixed_grp(int32_t qlen, int32_t tlen, int32_t init_score, int32_t ql, int32_t tl, int32_t W,
		int32_t M, int32_t X, int32_t I, int32_t D, int32_t E, int32_t T, const uint32_t *q32, int32_t qo, const uint32_t *t32, int32_t to,
		uint8_t *ztr, uint8_t *stage, uint32_t *runs, uint32_t *n_runs, unsigned long long *cells){
	static_assert(C >= 2 && C <= 8, "a lane's trace bytes of a row pair are one 64-bit word");
	const int gl = (int)(threadIdx.x & (WTZ_GRP_LANES - 1));
	constexpr int KB = 7;                      /* band column (< 128) inside the packed arg-max key; callers check |h| < 2^23 */
	wtz_aln_t x; memset(&x, 0, sizeof x);
	*n_runs = 0;
	if(init_score < 0) init_score = 0;
	int32_t hp[C], ep[C]; uint32_t nibp[C];
	const int32_t colrel0 = gl * C;
	/* "row -1": H(-1, j) = init_score + D + E*(j+1) (rh[] of kswx.h:257-259 one column to the left), E = -10000: row 0 is then an
	 * ordinary row whose band has not moved, with H(-1,-1) = init_score as the left boundary */
	#pragma unroll
	for(int k = 0; k < C; k++){ hp[k] = init_score + D + E * (colrel0 + k + 1); ep[k] = -10000; nibp[k] = 0; }
	int32_t mx = init_score, mi = -1, mj = -1, gmax = 0, gi = -1, gj = -1;
	int32_t jbp = 0, i, i_done = -1;
	unsigned long long ncell = 0;
	const int32_t CE = C * E, IE = I + E, DE = D + E;
	uint32_t tw = 0; int32_t tw_left = 0; uint32_t qcur = 0;
	WTZ_GLOBAL_AS unsigned long long *zg = wtz_as_global((unsigned long long*)ztr);
	for(i = 0; i < ql; i++){
		int32_t jb = i - W; if(jb < 0) jb = 0;
		int32_t je = i + W + 1; if(je > tl) je = tl;
		if((i & 15) == 0) qcur = wtz_lds16(q32, qo + i);
		const uint32_t qbase = (qcur >> ((i & 15) * 2)) & 3u;
		const int32_t j0 = jb + colrel0;
		const bool moved = (i > 0) && (jb != jbp);
		if(moved){ tw >>= 2; tw_left--; }
		if(i == 0 || tw_left < C){
			const int32_t jj = j0 < tl ? j0 : (tl > 0 ? tl - 1 : 0);
			tw = wtz_lds16(t32, to + jj);
			tw_left = 16;
		}
		/* ---- predecessors from the previous row's registers: the band either stays (H(i-1,j-1) is the previous column) or moves right
		 * by one (H(i-1,j-1) is the lane's own column, E(i-1,j) the next one).  Both hand-overs are fetched and selected: `moved` is
		 * uniform in a group but not across the groups of the wave ---- */
		const int32_t bnd = (i == 0) ? init_score : init_score + I + E * i;              /* H(i-1,-1), kswx.h:262 */
		int32_t prv = wtz_grp_from_prev(-10000, hp[C - 1]);
		prv = (gl == 0) ? bnd : prv;
		const int32_t nxt = wtz_grp_from_next(-10000, ep[0]);
		int32_t pred[C], ein[C];
		#pragma unroll
		for(int k = 0; k < C; k++){
			const int32_t stay_h = k ? hp[k - 1] : prv, move_e = (k + 1 < C) ? ep[k + 1] : nxt;
			pred[k] = moved ? hp[k] : stay_h;
			ein[k] = moved ? move_e : ep[k];
		}
		/* ---- m and the lane's F aggregate ---- */
		int32_t mv[C]; bool valid[C]; int32_t agg = -0x3FFFFFFF;
		#pragma unroll
		for(int k = 0; k < C; k++){
			const uint32_t tbase = (tw >> (2 * k)) & 3u;
			mv[k] = pred[k] + ((qbase == tbase) ? M : X);
			valid[k] = (j0 + k < je);
			const int32_t cand = mv[k] + DE + (C - 1 - k) * E;
			agg = (valid[k] && cand > agg) ? cand : agg;
		}
		int32_t f;
		{
			const int32_t g = agg - gl * CE;
			const int32_t pm = wtz_grp_max_scan_excl(g, -0x3FFFFFFF);
			const int32_t from_prev = (gl == 0) ? -0x3FFFFFFF : pm + (gl - 1) * CE;
			const int32_t from_init = -10000 + gl * CE;
			f = from_prev > from_init ? from_prev : from_init;
		}
		/* ---- H, E', F, trace nibble ---- */
		int32_t key = (int32_t)0x80000000;
		unsigned long long zb8 = 0;
		#pragma unroll
		for(int k = 0; k < C; k++){
			const int32_t m = mv[k], e = ein[k];
			int32_t h = m > e ? m : e;
			uint32_t nib = (m >= e) ? 0u : 1u;
			nib = (h < f) ? 2u : nib;
			h = h > f ? h : f;
			const int32_t te = m + IE, e2 = e + E;
			nib |= (e2 > te) ? 4u : 0u;
			const int32_t en = e2 > te ? e2 : te;
			const int32_t tf = m + DE, f2 = f + E;
			nib |= (f2 > tf) ? 8u : 0u;
			f = f2 > tf ? f2 : tf;
			hp[k] = valid[k] ? h : -10000; ep[k] = valid[k] ? en : -10000;
			nib = valid[k] ? nib : 0u;
			const int32_t kk = h * (1 << KB) + (colrel0 + k);
			key = (valid[k] && kk > key) ? kk : key;
			zb8 |= (unsigned long long)(nibp[k] | (nib << 4)) << (8 * k);
			nibp[k] = (i & 1) ? 0u : nib;
		}
		if(i & 1) zg[(size_t)(i >> 1) * WTZ_GRP_LANES + gl] = zb8;          /* both rows of the pair: 128 contiguous bytes per group */
		i_done = i;
		ncell += (unsigned long long)(je - jb);
		key = wtz_grp_max_all(key);
		int32_t imax = 0, mj2 = -1;
		if((key >> KB) >= 0){ imax = key >> KB; mj2 = jb + (key & ((1 << KB) - 1)); }           /* last j with h >= running max >= 0, kswx.h:288-289 */
		if(je == tlen){
			const int32_t idx = je - 1 - jb, kl = idx % C;
			int32_t hsel = hp[0];
			#pragma unroll
			for(int k = 1; k < C; k++) hsel = (kl == k) ? hp[k] : hsel;
			const int32_t h1 = wtz_grp_lane(hsel, idx / C);                  /* H(i, je-1) */
			if(gmax < h1){ gmax = h1; gi = i; gj = je - 1; }
		}
		if(i + 1 == qlen && gmax < imax){ gmax = imax; gi = i; gj = mj2; }
		jbp = jb;
		if(imax > mx){ mx = imax; mi = i; mj = mj2; }
		else if(imax <= 0) break;
	}
	if(i_done >= 0 && !(i_done & 1)){          /* the last row was the first of its byte pair */
		unsigned long long zb8 = 0;
		#pragma unroll
		for(int k = 0; k < C; k++) zb8 |= (unsigned long long)nibp[k] << (8 * k);
		zg[(size_t)(i_done >> 1) * WTZ_GRP_LANES + gl] = zb8;
	}
	if(cells && gl == 0) *cells += ncell;
	if(gmax > 0 && gmax >= mx + T){ x.score = gmax; x.qe = gi; x.te = gj; }
	else { x.score = mx; x.qe = mi; x.te = mj; }
	__threadfence_block();
	/* ---- traceback: the group stages blocks of 16 row pairs (2 KB) of the trace in LDS, its lane 0 walks there ---- */
	{
		int32_t i_ = x.qe, j_ = x.te; uint32_t d_ = 0;
		uint32_t run_op = 0xFFu, run_len = 0, nr = 0;
		uint32_t *stage32 = (uint32_t*)stage;
		constexpr int32_t RB = WTZ_GRP_STAGE_BYTES / WTZ_GRP_ZROW;
		while(i_ >= 0 && j_ >= 0){
			const int32_t p1 = i_ >> 1, p0 = p1 >= RB - 1 ? p1 - (RB - 1) : 0;
			{
				const uint32_t nd = (uint32_t)(p1 - p0 + 1) * (WTZ_GRP_ZROW >> 2);
				const uint32_t *src = (const uint32_t*)(ztr + (size_t)p0 * WTZ_GRP_ZROW);
				for(uint32_t xw = (uint32_t)gl; xw < nd; xw += WTZ_GRP_LANES) stage32[xw] = src[xw];
			}
			__threadfence_block();
			if(gl == 0){
				while(i_ >= 0 && j_ >= 0 && (i_ >> 1) >= p0){
					const int32_t col = j_ - (i_ > W ? i_ - W : 0);
					const uint32_t zv = stage[(size_t)((i_ >> 1) - p0) * WTZ_GRP_ZROW + (col / C) * 8 + (col % C)];
					const uint32_t nib = (zv >> ((i_ & 1) * 4)) & 0xFu;
					if(d_ == 0) d_ = nib & 3u; else if(d_ == 1) d_ = (nib & 4u) ? 1u : 0u; else d_ = (nib & 8u) ? 2u : 0u;
					if(d_ == 0){
						const int32_t pq = qo + i_, pt = to + j_;
						const uint32_t qb = (q32[pq >> 4] >> ((pq & 15) * 2)) & 3u, tq = (t32[pt >> 4] >> ((pt & 15) * 2)) & 3u;
						if(qb == tq) x.mat++; else x.mis++;
						i_--; j_--;
					}
					else if(d_ == 1){ i_--; x.ins++; }
					else { j_--; x.del++; }
					if(d_ == run_op) run_len++;
					else { if(run_len) runs[nr++] = (run_len << 4) | run_op; run_op = d_; run_len = 1; }
				}
			}
			i_ = wtz_grp_lane(i_, 0); j_ = wtz_grp_lane(j_, 0);
			__threadfence_block();
		}
		if(gl == 0){
			/* the two leading gaps (kswx.h:321-322) merge with the open run when the operation agrees (kswx_push_cigar) */
			if(i_ >= 0){ x.ins += i_ + 1; if(run_len && run_op == 1u) run_len += (uint32_t)(i_ + 1); else { if(run_len) runs[nr++] = (run_len << 4) | run_op; run_op = 1u; run_len = (uint32_t)(i_ + 1); } }
			if(j_ >= 0){ x.del += j_ + 1; if(run_len && run_op == 2u) run_len += (uint32_t)(j_ + 1); else { if(run_len) runs[nr++] = (run_len << 4) | run_op; run_op = 2u; run_len = (uint32_t)(j_ + 1); } }
			if(run_len) runs[nr++] = (run_len << 4) | run_op;
			*n_runs = nr;                              /* in traceback order: the caller replays them backwards */
			x.aln = x.mat + x.mis + x.ins + x.del; x.qe++; x.te++;
		}
	}
	return x;
}

/*
 * One window on one group.  Returns with *defer = true (nothing written) when the window has a problem outside the envelope
 * of the group form; *ok = false when the pool ran dry.  The result x is complete on lane 0 of the group.
 */
WTZ_D wtz_aln_t wtz_align_window_grp(const wtz_readview &pb1, const wtz_readview &pb2, const wtz_win_t &win, const wtz_zhit_t *anchors,
		wtz_cigar_t &cigar, const wtz_params_t *P, wtz_pool_t *pool, uint8_t *lds, unsigned long long *cells, bool *ok, bool *defer){
	const int gl = (int)(threadIdx.x & (WTZ_GRP_LANES - 1));
	const int32_t M = P->M, X = P->X, I = P->O, D = P->O, E = P->E, T = P->T;
	uint32_t *t32 = (uint32_t*)lds, *q32 = t32 + WTZ_GRP_SEQ_U32;
	uint8_t *stage = (uint8_t*)(q32 + WTZ_GRP_SEQ_U32);
	uint32_t *runs = (uint32_t*)(stage + WTZ_GRP_STAGE_BYTES);
	wtz_aln_t x, y; memset(&x, 0, sizeof x);
	*ok = true; *defer = false;
	const uint32_t a0 = win.anchors[0], a1 = win.anchors[1];
	if(a1 <= a0) return x;
	/* ---- the spans of both reads the window touches (its anchors bound every problem and every z-mer) ---- */
	int32_t t0 = 0x7FFFFFFF, t1 = 0, q0 = 0x7FFFFFFF, q1 = 0;
	for(uint32_t i = a0 + (uint32_t)gl; i < a1; i += WTZ_GRP_LANES){
		const wtz_zhit_t p = anchors[i];
		const int32_t o1 = (int32_t)ZH_OFF1(p), o2 = (int32_t)ZH_OFF2(p);
		t0 = o1 < t0 ? o1 : t0; q0 = o2 < q0 ? o2 : q0;
		t1 = o1 + (int32_t)ZH_LEN1(p) > t1 ? o1 + (int32_t)ZH_LEN1(p) : t1; q1 = o2 + (int32_t)ZH_LEN2(p) > q1 ? o2 + (int32_t)ZH_LEN2(p) : q1;
	}
	t0 = wtz_grp_min_all(t0); q0 = wtz_grp_min_all(q0); t1 = wtz_grp_max_all(t1); q1 = wtz_grp_max_all(q1);
	const int32_t span1 = t1 - t0, span2 = q1 - q0;
	if(span1 > WTZ_GRP_MAXSPAN || span2 > WTZ_GRP_MAXSPAN || t1 > (int32_t)pb1.len || q1 > (int32_t)pb2.len){ *defer = true; return x; }
	{
		const wtz_seq_packed s1 = pb1.sub(t0, 1), s2 = pb2.sub(q0, 1);
		unsigned long long *t64 = (unsigned long long*)t32, *q64 = (unsigned long long*)q32;
		for(int32_t w = gl; w * 2 < WTZ_GRP_SEQ_U32 - 1; w += WTZ_GRP_LANES){
			t64[w] = w * 32 < span1 ? wtz_pack32(s1, w * 32, span1) : 0ull;
			q64[w] = w * 32 < span2 ? wtz_pack32(s2, w * 32, span2) : 0ull;
		}
		if(gl == 0){ t32[WTZ_GRP_SEQ_U32 - 1] = 0; q32[WTZ_GRP_SEQ_U32 - 1] = 0; }
	}
	__threadfence_block();
	wtz_cigw_t Wc; Wc.v = &cigar; Wc.tail = 0;
	uint8_t *ztr = NULL; int32_t ztr_pairs = 0;         /* trace storage of the window, grown when a problem needs more row pairs */
	int32_t stop = 0;
	for(uint32_t i = a0; i < a1 && !stop; i++){
		const wtz_zhit_t p = anchors[i];
		const int32_t off1 = (int32_t)ZH_OFF1(p), off2 = (int32_t)ZH_OFF2(p);
		if(x.aln == 0){ x.tb = x.te = off1; x.qb = x.qe = off2; }
		if(off1 < x.te) continue;
		if(off2 < x.qe) continue;
		const int32_t qlen = off2 - x.qe, tlen = off1 - x.te;
		uint32_t n_runs = 0;
		if(qlen > 0 && tlen > 0){
			int32_t init = x.score < 0 ? 0 : x.score, W = P->w, ql = 0, tl = 0, n_col = 0;
			wtz_ext_geometry(qlen, tlen, init, W, M, I, D, E, T, ql, tl, n_col);
			const int32_t hmax = init + M * (ql < tl ? ql : tl);
			if(n_col > WTZ_GRP_LANES * WTZ_GRP_C || ql + tl + 4 > WTZ_GRP_RUNS || hmax >= (1 << 23)){ *defer = true; return x; }
			const int32_t need = (ql + 1) / 2;
			if(need > ztr_pairs){
				int32_t cap = ztr_pairs ? ztr_pairs : 32; while(cap < need) cap <<= 1;
				unsigned long long za = 0;
				if(gl == 0) za = (unsigned long long)(uintptr_t)wtz_pool_alloc(pool, (size_t)cap * WTZ_GRP_ZROW);
				za = wtz_grp_lane64(za, 0);
				if(za == 0){ *ok = false; return x; }
				ztr = (uint8_t*)(uintptr_t)za; ztr_pairs = cap;
			}
			y = wtz_extend_fixed_grp<WTZ_GRP_C>(qlen, tlen, x.score, ql, tl, W, M, X, I, D, E, T, q32, x.qe - q0, t32, x.te - t0, ztr, stage, runs, &n_runs, cells);
		} else {
			/* empty problem (wtz_extend_fixed: score = init, nothing aligned, empty CIGAR) */
			memset(&y, 0, sizeof y); y.score = x.score < 0 ? 0 : x.score;
		}
		if(gl == 0){
			x.score = y.score;
			x.aln += y.aln; x.mat += y.mat; x.mis += y.mis; x.ins += y.ins; x.del += y.del;
			x.te += y.te; x.qe += y.qe;
			for(uint32_t k = n_runs; k-- > 0;){ const uint32_t r = runs[k]; wtz_cigw_push(Wc, r & 0xFu, r >> 4); }
			if(x.te < off1){ x.del += off1 - x.te; x.aln += off1 - x.te; wtz_cigw_push(Wc, 2, (uint32_t)(off1 - x.te)); x.te = off1; }
			if(x.qe < off2){ x.ins += off2 - x.qe; x.aln += off2 - x.qe; wtz_cigw_push(Wc, 1, (uint32_t)(off2 - x.qe)); x.qe = off2; }
			const uint32_t len1 = ZH_LEN1(p), len2 = ZH_LEN2(p);
			/* one pass that writes its runs; a z-mer pair that turns out not to align (aln == 0) is rolled back: the writer's
			 * state is its open run plus the vector length */
			const uint32_t keep_tail = Wc.tail, keep_n = cigar.n;
			if(len1 <= 64 && len2 <= 64){
				wtz_seq_reg2 r1, r2;
				const int32_t p1 = off1 - t0, p2 = off2 - q0;
				r1.w0 = (uint64_t)wtz_lds16(t32, p1) | ((uint64_t)wtz_lds16(t32, p1 + 16) << 32); r1.w1 = (uint64_t)wtz_lds16(t32, p1 + 32) | ((uint64_t)wtz_lds16(t32, p1 + 48) << 32);
				r2.w0 = (uint64_t)wtz_lds16(q32, p2) | ((uint64_t)wtz_lds16(q32, p2 + 16) << 32); r2.w1 = (uint64_t)wtz_lds16(q32, p2 + 32) | ((uint64_t)wtz_lds16(q32, p2 + 48) << 32);
				y = wtz_align_zmer_w(r1, len1, r2, len2, M, I, D, E, &Wc);
			} else {
				wtz_seq_lds z1, z2; z1.s32 = t32; z1.off = off1 - t0; z2.s32 = q32; z2.off = off2 - q0;
				y = wtz_align_zmer_w(z1, len1, z2, len2, M, I, D, E, &Wc);
			}
			if(y.aln == 0){ Wc.tail = keep_tail; cigar.n = keep_n; stop = 1; }
			else {
				x.score += y.score;
				x.aln += y.aln; x.mat += y.mat; x.mis += y.mis; x.ins += y.ins; x.del += y.del;
				x.te += y.te; x.qe += y.qe;
			}
		}
		/* the lanes of the group follow the anchor loop with the same position, score and stop decision */
		x.score = wtz_grp_lane(x.score, 0); x.te = wtz_grp_lane(x.te, 0); x.qe = wtz_grp_lane(x.qe, 0); x.aln = wtz_grp_lane(x.aln, 0);
		stop = wtz_grp_lane(stop, 0);
	}
	if(gl == 0) wtz_cigw_finish(Wc);
	return x;
}

#endif /* __HIPCC__ */
#endif
