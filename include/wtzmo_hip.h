/*
 * wtzmo_hip.h — C ABI of libwtzmo_hip.so, the MI355X (gfx950) implementation of SMARTdenovo's
 * wtzmo hot path.  Plain C, caller-owned host buffers, opaque device context, no callbacks.
 *
 * The reference has no plugin/FFI boundary: the path's boundary is the `wtzmo` process itself
 * (SURVEY.md §8b).  This ABI therefore mirrors the reference's *function* granularity, so that each
 * entry point can be parity-tested against the reference function it replaces, and is consumed by
 * our own host driver (smartdenovo_amd/csrc/host/, the drop-in `wtzmo` executable):
 *
 *   wtz_upload_reads     <- push_long_read_wtzmo / BaseBank          wtzmo.c:207-215, dna.h:318-410
 *   wtz_index_build      <- index_wtzmo                               wtzmo.c:349-430
 *   wtz_zindex_build     <- index_single_read_seeds (all reads once)  hzm_aln.h:70-115
 *   wtz_candidates       <- query_wtzmo                               wtzmo.c:433-573
 *   wtz_pairs_seed       <- query_single_read_seeds + process_hzmps + merge_paired_kmers_window +
 *                           chaining_wtseedv | dot_matrix_align_hzmps hzm_aln.h:173-224, 580-713, 1134-1186
 *   wtz_pairs_align      <- fast_seeds_align_hzmo + global_align_regs_hzmo (kswx_extend_align_core,
 *                           kswx_extend_align_shift_core, ksw_global2) hzm_aln.h:1247-1486, kswx.h:101-335, ksw.c:503-586;
 *                           with params.refine also kswx_refine_alignment kswx.h:483-659
 *
 *   wtz_pairs_seed + wtz_pairs_align with params.aux_strand = 1
 *                        <- align_hzmaux, the pair routine of wtgbo   hzm_aln.h:1684-1775 (its caller wtgbo.c:37-56 orients the read first:
 *                           the host uploads every read and its reverse complement, smartdenovo_amd/csrc/host/wtgbo_main.c)
 *
 * Everything returned is integer and bit-exact against `wtzmo -t 1`.  The order-dependent state of
 * the reference (closed pairs, contained-read masking, per-read coverage: wtzmo.c:806-822, 1065-1100,
 * 1309-1334) stays with the caller: these functions are pure in (reads, parameters, arguments).
 *
 * Threading: one context per GPU, thread-compatible (not thread-safe per context).
 * Errors: 0 on success, a negative WTZ_E_* otherwise; wtz_last_error() has the message.
 */
#ifndef WTZMO_HIP_H
#define WTZMO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WTZ_OK        0
#define WTZ_E_ARG    -1   /* bad argument */
#define WTZ_E_HIP    -2   /* HIP runtime error (no device, launch failure, ...) */
#define WTZ_E_POOL   -3   /* device scratch pool exhausted: retry with fewer items or a larger pool */
#define WTZ_E_STATE  -4   /* call order violated (e.g. align before seed) */

/* Parameters: one field per reference WTZMO field that the path reads (wtzmo.c:98-132, 1663-1689). */
typedef struct {
	uint32_t ksize, zsize, hk, hz, ksave, kovl, ncand, nbest;
	uint32_t kwin, kstep, ztot, zovl, max_kmer_freq, max_zmer_freq, max_kmer_var;
	float    win_rep_norm, win_rep_cutoff;
	int32_t  w, ew, W, M, X, O, E, T;
	int32_t  min_score; float min_id;
	int32_t  dot_matrix, xvar, yvar, min_block_len, max_overhang;
	float    deviation_penalty, gap_penalty;
	int32_t  refine;         /* -n: kswx_refine_alignment after stitching (wtzmo.c:1031-1034) */
	int32_t  aux_strand;     /* 1 = the pair stages in align_hzmaux's form (hzm_aln.h:1693-1699, wtgbo's caller): only same-strand z-mer matches are kept
	                          * (filter_by_region_hzmps with dir 0, hzm_aln.h:1188-1197, before the window merge), there is no n_hits gate, and the chain
	                          * of strand 0 is kept when its weight is >= ztot (the caller passes -R there: hzm_aln.h:1699 compares with zovl) */
} wtz_params_c;

typedef struct wtz_ctx wtz_ctx_t;

typedef struct {
	uint64_t n_occ;          /* sampled k-mer occurrences in the indexed range */
	uint64_t n_distinct;     /* distinct sampled k-mers (ktyp) */
	uint64_t ktot;           /* sum of saturating counts (wtzmo.c:380-388) */
	uint64_t n_kept;         /* k-mers with 2 <= cnt <= K (the hash table content) */
	uint32_t max_kmer_freq;  /* resolved K */
	uint32_t avg_rdlen;      /* wtzmo.c:361-368 */
} wtz_index_stats_t;

typedef struct {
	uint32_t n_hits;         /* |cache| of the pair */
	uint32_t gate;           /* n_hits*zsize >= ztot (wtzmo.c:857) */
	uint32_t ovl[2];         /* zmo: SEED[dir].ovl after chaining_wtseedv (0 when no window) */
	uint32_t nwin[2];        /* zmo: chain windows kept for strand dir (only when ovl >= ztot) */
	int32_t  dm_score, dm_qb, dm_qe, dm_tb, dm_te, dm_dir;   /* dmo: kswr_t of dot_matrix_align_hzmps */
} wtz_pair_summary_t;

typedef struct { int32_t beg[2], end[2]; } wtz_winbox_t;     /* wt_seed_t.beg/end of a chain window */

typedef struct {
	int32_t  score, tb, te, qb, qe, aln, mat, mis, ins, del;   /* kswx_t of global_align_regs_hzmo */
	uint32_t n_regs;         /* windows that passed wtzmo.c:1026; 0 means "regs->size == 0" (wtzmo.c:1029) */
	uint32_t cigar_len;      /* number of (len<<4|op) words, op M0/I1/D2 */
	uint64_t cigar_off;      /* offset of this item's CIGAR inside the buffer returned by wtz_fetch_cigars */
	uint32_t text_len;       /* length of the CIGAR as text ("%d[MID]"..., kswx_cigar2string kswx.h:1093-1120), no terminator */
	uint32_t pad;
	uint64_t text_off;       /* offset inside the buffer returned by wtz_fetch_cigar_text */
} wtz_aln_result_t;

typedef struct {
	double   ms_index, ms_zindex, ms_candidates, ms_pairs, ms_winalign, ms_stitch;   /* HIP-event kernel time per stage */
	uint64_t n_candidates_q, n_pairs, n_winalign, n_stitch;                           /* work items launched */
	uint64_t cells_shift, cells_fixed, cells_global;                                  /* DP cell updates as the reference loops execute them */
	uint64_t bytes_seed_algo;                                                         /* algorithmic bytes of seed lookup */
	uint64_t pool_peak;
	double   ms_ext;                                                                  /* K-sw3 wave kernel alone (inside ms_stitch) */
	uint64_t n_extjobs;
	double   ms_gap;                                                                  /* K-sw2 gap kernels alone (inside ms_stitch) */
	uint64_t bytes_zmer_algo;                                                         /* algorithmic bytes of z-mer matching: candidate L/4 + 16 B per emitted match (hzm_aln.h:173-224) */
	double   ms_ingest;                                                               /* K_pack_ascii + K_pack_fix of wtz_upload_reads_ascii (kernel time, copies excluded) */
	uint64_t bytes_ingest_algo;                                                       /* 1 byte read + 2 bits written per base */
} wtz_counters_t;

const char *wtz_last_error(void);
int  wtz_device_count(void);
/* free / total bytes of the device's HBM right now.  No reference counterpart: the reference sizes nothing (its indexes live in host RAM, wtzmo.c:249-430,
 * hzm_aln.h:70-115); the host driver sizes the scratch pool and chooses between the all-reads z-mer index and the per-batch one from it. */
int  wtz_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes);

int  wtz_ctx_create(int device, const wtz_params_c *params, uint64_t pool_bytes, wtz_ctx_t **out);
void wtz_ctx_destroy(wtz_ctx_t *ctx);
/* second context on the same GPU sharing the parent's reads + indexes (read-only), with its own HIP stream, scratch pool and
 * batch state: one per host thread that keeps a batch in flight. pool_bytes 0 = same as the parent. */
int  wtz_ctx_clone(wtz_ctx_t *parent, uint64_t pool_bytes, wtz_ctx_t **out);

/* bits: 2-bit packed bases, 32 per word, base i at bits ((~i)&31)*2 of word i>>5 (dna.h:78);
 * rdoff/rdlen per read in READ-ID order (id = rank by length DESC under the reference's sort, wtzmo.c:1708). */
int  wtz_upload_reads(wtz_ctx_t *ctx, const uint64_t *bits, uint64_t n_words, const uint64_t *rdoff, const uint32_t *rdlen, uint32_t n_reads);

/* f4: the same upload from the bases as TEXT - seq2basebank (dna.h:397-410) on the device.  seq = the sequences of all reads of the
 * input concatenated in FILE order (n_bases bytes; upper or lower case; rdoff / rdlen index into it exactly as they would into the packed bank).
 * Every byte that is not one of ACGTacgt becomes `lrand48() & 3`, drawn in file order like the reference's loader does (dna.h:405) with the state of a
 * glibc process that never called srand48 (X0 = 0); rand_calls_before = lrand48 calls the emulated process made before this input (0 for wtzmo / wtgbo), *n_random (may be NULL) = such
 * bytes found.  The resulting device bank is bit-identical to wtz_upload_reads of the host-packed bank. */
int  wtz_upload_reads_ascii(wtz_ctx_t *ctx, const char *seq, uint64_t n_bases, const uint64_t *rdoff, const uint32_t *rdlen, uint32_t n_reads, uint64_t rand_calls_before, uint64_t *n_random);
/* wtgbo's view of the reads (wtgbo.c:48-49 reverse-complements a '-' candidate BEFORE its z-mers are taken; revbitseq_basebank, dna.h): appends to the n uploaded
 * reads their reverse complements as reads n .. 2n-1 (read n + i = reverse complement of read i over its current [rdoff, rdoff + rdlen) range), packed on the device.
 * Indexes built before the call are dropped. */
int  wtz_append_revcomp_views(wtz_ctx_t *ctx);
/* the packed bank back from the device (n_words = (n_bases + 31) / 32): what a host-side consumer of the 2-bit reads (wtgbo's reverse-complement views, tests) reads */
int  wtz_fetch_read_bits(wtz_ctx_t *ctx, uint64_t *bits, uint64_t n_words);

/* A2 over read ids [id_beg, id_end). *max_kmer_freq: in = -K (0/1 = auto), out = resolved cutoff. */
int  wtz_index_build(wtz_ctx_t *ctx, uint32_t id_beg, uint32_t id_end, uint32_t *max_kmer_freq, wtz_index_stats_t *stats);

/* ---- k-mer index SHARDED by read-id range over several contexts (BASELINE configs[4]; the reference's -G splits the index the
 * same way, wtzmo.c:1281-1285, but filters each part by its own counts).  To equal the UNSHARDED output the frequency filter (cnt > K or
 * cnt <= 1, wtzmo.c:401-405) and the automatic cutoff (wtzmo.c:380-393) must see the counts of all shards:
 *   wtz_index_count        occurrences of the shard's reads [id_beg, id_end), ordered by k-mer; *n_distinct = its distinct sampled k-mers
 *   wtz_index_counts_fetch those k-mers (ascending) with their occurrence counts in this shard
 *   ... the caller adds the counts of equal k-mers over the shards (one exchange), derives K ...
 *   wtz_index_finish       total_cnt[i] = count of the i-th fetched k-mer over ALL shards; builds the shard's table of the k-mers with
 *                          2 <= min(total, 0xFFFF) <= K, each pointing at the shard's own seed run.
 * A query is then answered by every shard (wtz_candidate_groups_*): the (read, strand) groups of a query with ol >= -d, as
 * `(read << 1 | strand) << 32 | ol` in key order.  Shards are contiguous id ranges, so the shards' lists concatenated in shard order are
 * the unsharded list, and wtz_cand_tail_host replays the strand merge and the candidate heap (wtzmo.c:516-571) over it. */
int  wtz_index_count(wtz_ctx_t *ctx, uint32_t id_beg, uint32_t id_end, uint64_t *n_distinct, uint64_t *n_occ);
int  wtz_index_counts_fetch(wtz_ctx_t *ctx, uint64_t *kmers, uint32_t *cnts);
int  wtz_index_finish(wtz_ctx_t *ctx, const uint32_t *total_cnt, uint32_t K, uint64_t *n_kept);
int  wtz_candidate_groups_begin(wtz_ctx_t *ctx, const uint32_t *qids, uint32_t nq);
int  wtz_candidate_groups_end(wtz_ctx_t *ctx, uint32_t *ngroups);                    /* ngroups[i] = groups of query i */
int  wtz_candidate_groups_fetch(wtz_ctx_t *ctx, uint64_t *groups, uint64_t total);   /* packed in query order; total = sum of ngroups */
void wtz_cand_tail_host(const uint64_t *groups, uint32_t ng, uint32_t kovl, uint32_t ncand, uint64_t *heap, uint32_t *hn);

/* A5 for every read + the candidate-side occurrence caps of A6. Call once after wtz_upload_reads. */
int  wtz_zindex_build(wtz_ctx_t *ctx);
/* The same index for the listed reads only (ascending ids); every other read gets an empty slice.  For read sets whose whole z-index
 * (16 B per base) does not fit beside the scratch pool: the caller rebuilds it per batch of queries from the batch's queries + candidates. */
int  wtz_zindex_build_subset(wtz_ctx_t *ctx, const uint32_t *ids, uint32_t n);
/* Several GPUs: the pairs of a batch are dealt by CANDIDATE id, so a device only ever walks the z-mers of its own share of the
 * reads as candidates - the index above then holds that share only (wtz_zindex_build_subset once) - but it needs the table of every QUERY of
 * the batch (hzm_aln.h:70-115 builds exactly that table per query, wtzmo.c:842-845).  This call builds a second, small index of the listed
 * reads (ascending ids) that the pair stages read the query side from; it is rebuilt per batch and dropped by any rebuild of the index above
 * or by ids == NULL, n == 0.  Results are identical to one index holding all reads. */
int  wtz_zindex_build_queries(wtz_ctx_t *ctx, const uint32_t *ids, uint32_t n);

/* A3. cand: nq rows of (ncand+1) u64 `id<<32|ol`; ncand_io[i]: in = entries already in row i
 * (candidate heaps carried across -G index parts, else 0), out = entries after this index part.
 * Rows are the reference's heap arrays verbatim (the caller applies wtzmo.c:813-822). */
int  wtz_candidates(wtz_ctx_t *ctx, const uint32_t *qids, uint32_t nq, uint64_t *cand, uint32_t *ncand_io);
/* the same request split in two, so that the caller's sequential work (the commit of the previous batch) overlaps the kernel:
 * _begin uploads and launches and returns at once; _end waits and fills cand / ncand_out (nq rows as passed to _begin).  No other
 * call on this context may be made in between. */
int  wtz_candidates_begin(wtz_ctx_t *ctx, const uint32_t *qids, uint32_t nq, const uint64_t *cand, const uint32_t *ncand_in);
int  wtz_candidates_end(wtz_ctx_t *ctx, uint64_t *cand, uint32_t *ncand_out);

/* Start a batch: releases all per-batch device results of the previous one. */
int  wtz_batch_begin(wtz_ctx_t *ctx);

/* A5/A6/A7 for n (query, candidate) pairs. Results stay on the device for wtz_pairs_windows / wtz_pairs_align. */
int  wtz_pairs_seed(wtz_ctx_t *ctx, const uint32_t *qid, const uint32_t *cid, uint32_t n, wtz_pair_summary_t *out);

/* Chain windows of the last wtz_pairs_seed, packed in (pair, strand) order: sum of nwin[] entries. */
int  wtz_pairs_windows(wtz_ctx_t *ctx, wtz_winbox_t *wins, uint64_t n_wins);

/* A9 + A10 for m (pair index into the last wtz_pairs_seed, strand) items. */
int  wtz_pairs_align(wtz_ctx_t *ctx, const uint32_t *pair_idx, const uint8_t *dir, uint32_t m, wtz_aln_result_t *out);
int  wtz_fetch_cigars(wtz_ctx_t *ctx, uint32_t *dst, uint64_t n_ops);
/* the same CIGARs already rendered as text on the device (what the .ovl column 17 holds): sum of text_len bytes */
int  wtz_fetch_cigar_text(wtz_ctx_t *ctx, char *dst, uint64_t n_bytes);
/* wtz_fetch_cigar_text in two halves (kswx_cigar2string kswx.h:1093-1120, as above): _begin renders the text and starts its copy to dst on a stream of its own and
 * returns, _end waits for that copy.  In between the context takes the next range's calls (wtz_batch_begin ... wtz_pairs_align): the copy - 3.4 GB per configs[2]
 * step, 72 ms at the rate of the link - runs beside their kernels.  dst (page-locked: wtz_host_alloc) must not be read before _end returns.  _end may be called
 * from another thread than the one that drives the context. */
int  wtz_fetch_cigar_text_begin(wtz_ctx_t *ctx, char *dst, uint64_t n_bytes);
int  wtz_fetch_cigar_text_end(wtz_ctx_t *ctx);
/* the same text left ON THE DEVICE: *dev_ptr = device address of the n_bytes (valid until the next call on this context).  For a rank that does not write
 * records itself (one process per GPU): the ~6 KB of CIGAR text per record go from this buffer to the committing rank's GPU over xGMI
 * (RCCL send of a tensor that aliases it) without passing through this rank's host memory. */
int  wtz_cigar_text_device(wtz_ctx_t *ctx, uint64_t n_bytes, void **dev_ptr);
/* Page-locked host memory for the buffers the library copies results into (the CIGAR text is ~6 KB per record: a pageable
 * destination halves the copy rate).  Plain malloc semantics otherwise; NULL on failure.  No reference counterpart: the
 * reference formats its records in the worker's own heap (wtzmo.c:1064-1095). */
void *wtz_host_alloc(uint64_t n_bytes);
void  wtz_host_free(void *p);

/* ---- TEST-ONLY entry: function-level parity of every device form of the three banded DPs against vectors dumped from the
 * reference's own routines (tests/golden/dp_vectors.npz, tests/test_gpu_dp_forms.py).  One DP problem per element on sequences taken
 * from the uploaded reads; nothing in the product path calls it.
 *   kind WTZ_DP_SHIFT  <- kswx_extend_align_shift_core kswx.h:101-232   (W as passed to it: negative = exact band)
 *        WTZ_DP_FIXED  <- kswx_extend_align_core       kswx.h:234-335   (band = params.w, as the path always calls it)
 *        WTZ_DP_GLOBAL <- ksw_global2                  ksw.c:503-586    (W = band width of this ONE call; gap costs from params)
 *   form 0 = the form the product would pick (for WTZ_DP_SHIFT: the whole run_extjobs dispatch); otherwise
 *        WTZ_DP_SHIFT : 1 one-wave register kernel, 2 four-wave kernel, 3 LDS-ring kernel (+ its scalar fallback), 4 scalar body
 *        WTZ_DP_FIXED / WTZ_DP_GLOBAL : C | 16*pool  (C = 1, 2, 4, 8 band columns per lane; +16 = 4-bit trace in the pool instead of LDS),
 *                       255 scalar body; WTZ_DP_GLOBAL also 32 = LDS-ring wave DP, 33 = the same with the 72 KB slice of the wide launch
 *   out[i].form_used = the form that produced the result, 0 = the forced form does not cover this problem (result untouched). */
#define WTZ_DP_SHIFT  0
#define WTZ_DP_FIXED  1
#define WTZ_DP_GLOBAL 2
typedef struct {
	uint32_t q_read, t_read;        /* read ids */
	uint32_t q_rev, t_rev;          /* 1 = reverse-complement view of the read (the candidate's strand '-') */
	int32_t  q_from, t_from;        /* logical position of base 0 of the problem inside the view */
	int32_t  q_strand, t_strand;    /* +1 / -1: walking direction from there (left extensions walk backwards) */
	int32_t  q_len, t_len, init_score, W;
} wtz_dp_problem_t;
typedef struct {
	int32_t  score, tb, te, qb, qe, aln, mat, mis, ins, del;      /* kswx_t; WTZ_DP_GLOBAL: score, aln, mat, mis, ins, del */
	uint32_t cigar_len, form_used;
	uint64_t cigar_off;             /* offset (in words) of this problem's CIGAR inside `cigar` */
	uint64_t cells;                 /* DP cells as the reference loops execute them (extensions) */
} wtz_dp_result_t;
int  wtz_test_dp(wtz_ctx_t *ctx, int32_t kind, int32_t form, const wtz_dp_problem_t *problems, uint32_t n, wtz_dp_result_t *out, uint32_t *cigar, uint64_t cigar_cap);

/* f2: kswx_extend_align (kswx.h:469-481 = kswx_extend_align_shift_core kswx.h:101-232, W as the caller passes it) for n independent problems on
 * views of the uploaded reads - what wtext's worker calls twice per overlap (wtext.c:250, 268) after it clipped the overlap's CIGAR to the retained regions.
 * The jobs go through the same dispatch as the end extensions of wtzmo's stitched alignments.  out[i] = the kswx_t of the call (score, qe, te, aln, mat, mis, ins, del;
 * an empty side returns the start score clamped at 0, like kswx.h:113-118) and cigar_off / cigar_len of its operations (len << 4 | op, first operation first) in `cigar`. */
int  wtz_extend_batch(wtz_ctx_t *ctx, const wtz_dp_problem_t *problems, uint32_t n, wtz_dp_result_t *out, uint32_t *cigar, uint64_t cigar_cap);

/* Scratch accounting, so that the caller can size its batches to the pool instead of discovering the limit by WTZ_E_POOL:
 * the context's scratch is cut into a main pool (everything that lives for the batch: match lists, windows, CIGARs) and a transient
 * pool (K-sw3 trace matrices; the library sizes its own launch groups to it).  main_used = bytes of the main pool in use after the
 * last stage call (wtz_pairs_seed / wtz_pairs_align), transient_peak = high-water mark of the transient pool during it. */
typedef struct { uint64_t main_cap, main_used, transient_cap, transient_peak; } wtz_pool_info_t;
int  wtz_pool_info(wtz_ctx_t *ctx, wtz_pool_info_t *out);
/* which pool the context's last WTZ_E_POOL came from: 0 = none yet, 1 = the main pool (the caller's bytes-per-pair estimate was too low: plan smaller ranges from
 * now on), 2 = the transient trace pool of the K-sw3 launches (the library's own trace budget was too low and has been raised: redo the range, keep the estimate) */
int  wtz_pool_failure_kind(wtz_ctx_t *ctx);

int  wtz_get_counters(wtz_ctx_t *ctx, wtz_counters_t *out);
int  wtz_reset_counters(wtz_ctx_t *ctx);

#ifdef __cplusplus
}
#endif
#endif
