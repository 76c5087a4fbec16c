/*
 * wtzmo — MI355X-native drop-in for SMARTdenovo's `wtzmo` (same argv, same .ovl / .contained files).
 *
 * Host side in plain C.  All hot-path computation (k-mer index + seed lookup, z-mer matching,
 * window detection / chaining or the dot-matrix engine, the three banded DPs) runs on the GPU
 * through the C ABI of libwtzmo_hip.so (include/wtzmo_hip.h).  What stays here is exactly the part
 * of the reference that is sequential by definition: the order-dependent commit of `wtzmo -t 1`
 * (closed pairs, contained-read masking, per-read coverage, record order; wtzmo.c:806-822, 933-986,
 * 1005-1123, 1170-1249, 1309-1334), replayed over device results that are pure functions of
 * (query, candidate).
 *
 * Execution shape: queries are taken in id order in batches; for a batch the GPU computes, for every
 * candidate pair not already closed when the batch is planned, the pair result and the alignment of
 * its best strand (a superset of what the reference would touch, because the closed/masked sets only
 * grow); the host then commits the batch strictly in query order.  Batch size adapts to the fraction
 * of speculative work that the commit discards.
 *
 * Extra options (not in the reference):  --gpu <id>, --pool-gb <n>, --batch <max queries>,
 * --stats <file> ("pairs\tpair_bp\toverlap_seconds\tindex_seconds"), --lib-check.
 */
#define _GNU_SOURCE
#include <getopt.h>
#include <unistd.h>
#include <time.h>
#include <pthread.h>
#include <sys/uio.h>
#include <errno.h>
#include <sys/stat.h>
#include <fcntl.h>
#include "wtz_host.h"
#include "wtz_ovlb.h"

static int usage(void){
	printf(
	"WTZMO (MI355X): overlapper of long reads using homopolymer compressed k-mer seeding\n"
	"Drop-in for SMARTdenovo wtzmo 1.0 -- identical options and file formats; hot path on the GPU\n"
	"Usage: wtzmo [options]\n"
	" -t <int>    accepted for compatibility (the GPU path is deterministic and equals `wtzmo -t 1`)\n"
	" -P <int> -p <int>  total parallel jobs / index of this job\n"
	" -i <string> long reads (+) *   -I <string> query-only reads (+)   -b <string> clip regions (+)\n"
	" -J <int>    min read length    -o <string> output *  -f overwrite   -9 <string> tested pairs out\n"
	" -L <string> tested pairs in (+)  -F <string> reads to mask (+)  -C no .contained file  -N seeds only\n"
	" -H <int> -k <int> -K <int> -S <int> -d <int> -G <int>      k-mer seeding [3,16,0,4,300,1]\n"
	" -z <int> -Z <int> -l <int> -y <int> -R <int> -r <int> -q <int>  z-mer windows [10,64,2,800,200,300,100]\n"
	" -U <float>x5 | -U -1   dot-matrix engine    -A <int> -B <int> candidates / best [500,100]\n"
	" -w -e -W -M -X -O -E -T  alignment [50,800,3200,2,-5,-3,-1,-50]   -s <int> -m <float> [200,0.5]\n"
	" --gpu <int> | --gpus <int> (one process, N GPUs, ONE .ovl identical to -t 1)  --pool-gb <int> --batch <int> --stats <file>\n"
	" --shard-index  --zindex-batch <0|1>  --binary-out (64-byte records, include/wtz_ovlb.h; read by wtgbo --binary-in / wtovl)\n");
	return 1;
}

typedef struct { char **a; int n, cap; } strlist_t;
static void sl_push(strlist_t *l, char *s){ if(l->n == l->cap){ l->cap = l->cap ? l->cap * 2 : 4; l->a = (char**)hx_realloc(l->a, sizeof(char*) * (size_t)l->cap); } l->a[l->n++] = s; }

typedef struct { uint64_t e; uint32_t pidx; uint32_t pad; } cand_t;
#define GT_CAND(a, b) ((uint32_t)(b)->e > (uint32_t)(a)->e)                  /* wtzmo.c:821 */
HX_DEFINE_SORT_EXACT(sort_cands_exact, cand_t, GT_CAND)
static int gt_read(const void *a, const void *b, void *ctx){ (void)ctx; return ((const hx_read_t*)b)->len > ((const hx_read_t*)a)->len; }               /* wtzmo.c:1708 */

typedef struct { uint32_t pb2, dir, ovl, closed, pidx; } seed_t;
#define GT_SEED(a, b) ((b)->ovl > (a)->ovl)                                 /* wtzmo.c:986 */
HX_DEFINE_SORT_EXACT(sort_seeds_exact, seed_t, GT_SEED)
/* what the order-free part of a query's commit (closed filter, candidate order, window depth, seed weights, seed order) leaves for the sequential part;
 * buffers grow and are kept.  dep: +1 / -1 marks, then the running depth, as 16-bit words (the reference's counters are u2i and wrap the same way): zero between queries */
typedef struct {
	cand_t *cand; size_t capcand; seed_t *seeds; size_t capseeds; uint16_t *dep; size_t capdep;
	uint32_t nc, nseed, ngate; double t[5];      /* t: candidate rows + closed filter + order | window marks | running depth | seed weights | seed order */
} cq_work_t;

typedef struct { uint32_t pb1, pb2, dir2; int qb, qe, tb, te, score, mat, mis, ins, del, aln; char *cigar; } hit_t;

typedef struct {      /* results of the query processed last, not yet merged into the global state */
	uint32_t rd_id;
	hit_t *hits; size_t nhit, caphit;
	uint32_t *masks; size_t nmask, capmask;
	uint64_t *closed; size_t nclosed, capclosed;
	seed_t *seeds; size_t nseed, capseed;
} pending_t;

typedef struct eng_s {
	wtz_params_c P; int do_align; uint32_t n_idx, n_job, i_job;
	hx_store_t st; uint8_t *masked; uint32_t *rdcovs; hx_set_t closed;
	uint32_t *rdlen; uint32_t avg_rdlen;
	uint64_t *closed_order; size_t n_order, cap_order; int keep_order;      /* -9: the pairs in the order they entered closed_alns (the file is a replay of it) */
	wtz_ctx_t *ctx; FILE *out;
	double ing_ms; uint64_t ing_bytes;      /* f4: kernel time / algorithmic bytes of the device ingest at load time (wtz_upload_reads_ascii) */
	int zbatch;                 /* --zindex-batch (automatic above ~2.4 Gbp of reads): the z-mer index is rebuilt per batch of queries for the batch's queries + candidates instead
	                             * of once for all reads (16 B per base: 160 GB at BASELINE configs[3]) */
	int binary_out;             /* --binary-out: 64-byte binary records behind a name table (include/wtz_ovlb.h) instead of the 17 text columns; the CIGAR text is not fetched */
	int zsplit;                 /* several parts (devices or ranks): part d's z-mer index holds the candidate side of the reads = d (mod nparts) only, the queries of a batch get their own small
	                             * index per batch (wtz_zindex_build_queries) - the replicated all-reads build (0.135 s of a 3.2 s configs[2] step, not divided by N) becomes 1 / N of it */
	int shard;                  /* --shard-index: the k-mer index is sharded by read-id range over the devices (reads and z-index stay replicated); output == unsharded */
	uint32_t ndev; int devs[8]; wtz_ctx_t *ctxs[8];      /* --gpus N / --gpu-list: one context per device, reads + both indexes replicated; ctx == ctxs[0] */
	uint64_t pair_bp, n_pairs, nrec;
	/* candidate rows carried across -G index parts (the reference's rdhits), else NULL */
	uint64_t *rows; uint32_t *nrow; uint32_t stride; int rows_all;
	uint32_t max_batch, first_batch, n_workers; int first_batch_set;
	/* pipeline: batches are planned in query order by whichever worker is free, computed on that worker's context
	 * (own HIP stream + scratch pool) and committed strictly in sequence */
	pthread_mutex_t mu; pthread_cond_t cv;
	uint32_t cursor, qend, B; uint64_t next_seq, commit_seq;
	/* planned batch sizing: main-pool bytes a pair has needed so far (largest seen, reads are processed longest first), so that the pairs of
	 * a range are cut to fit the pool BEFORE the device stages run; WTZ_E_POOL and the halving below it remain as the safety net */
	double bytes_per_pair, bpp_decay; uint64_t main_cap; uint64_t n_split, n_ranges;
	pending_t pend;
	/* commit_query scratch kept across queries (round 6: the two calloc'ed depth arrays of a query were 60 KB of zeroing + a 10 000-step scalar prefix each -
	 * more than half of the sequential commit): cq_dep = +1 / -1 marks, then the running depth, as 16-bit words (the reference's counters are u2i and wrap the
	 * same way); only the span the query's windows touch is summed, and it is zeroed again behind the query */
	cq_work_t cq;                      /* the commit thread's own */
	struct cq_spec_s *spec;            /* helper threads that prepare the queries ahead of the commit (cq_spec_*) */
	uint32_t *closed_touch;            /* per read: number of closed pairs with this read added so far; a prepared query is valid iff its read's count has not moved */
	uint64_t n_spec_hit, n_spec_miss; double t_spec_wait, t_spec_ctl, t_flush;
	/* stats */
	char *cig_keep[16]; uint64_t cig_keep_cap[16];      /* page-locked CIGAR text buffer of worker w, kept across steps (pinning is the expensive part) */
	double t_cq1[4];            /* section 1 of commit_query in its parts: window marks | running depth | seed weights | seed order */
	double t_cq[4];             /* commit_query sections: candidate rows + closed filter + sort | window depth + seed weights + sort | hits (gates, queueing, masking) | plan_pairs */
	double t_gpu, t_commit, t_zbatch, t_call[6], t_io[2];      /* t_io: waiting for the writer thread before a text buffer is reused / at the end of the run */      /* t_call: wall seconds inside wtz_candidates / pairs_seed / pairs_windows / pairs_align / fetch_cigar_text / planning */
	uint64_t spec_pairs, used_pairs, spec_items, used_items, spec_queries, used_queries, n_batches;
	double pair_row_ratio;       /* planned pairs per candidate row of the ranges before (plan_pairs): the rows bound a range, closed pairs and masked queries thin them out */
	uint64_t n_masked;           /* reads masked by commits of this step (flush_pending) */
	double cand_bpq; uint64_t cand_main_cap; uint32_t cand_cap0;      /* seed lookup: main-pool bytes per query of the requests so far (the largest), the pool they ran in; cand_cap0: batch cap until the first measurement (large inputs) */
	double last_mask_rate;       /* new masks per used query of the batch that finished last (process_batch: may the next batch be formed in front of the last commit?) */
	double extra_ms[6]; uint64_t extra_u64[7];      /* counters of the cloned contexts */
} eng_t;

/* ---- ranks: with one process per GPU (torchrun; bench.py / tests set the exchange hooks through wtzmo_set_dist) the parts of rank 0's
 * batches live in the OTHER PROCESSES: rank 0 plans and commits (the order-dependent part of `wtzmo -t 1` is one sequential
 * stream by definition), every rank runs the pure device stages of its share of the pairs / candidate requests on its own GPU with
 * reads and both indexes replicated, and the results travel to rank 0 (RCCL send / recv when the hooks sit on the nccl backend).
 * One .ovl, written by rank 0, identical to `wtzmo -t 1` for any number of ranks; total work is fixed (strong scaling). ---- */
typedef void (*wtz_dist_bcast_fn)(void *buf, uint64_t nbytes);                   /* from rank 0, same nbytes on every rank */
typedef void (*wtz_dist_send_fn)(const void *buf, uint64_t nbytes, int dst);
typedef void (*wtz_dist_recv_fn)(void *buf, uint64_t nbytes, int src);
typedef void (*wtz_dist_send_dev_fn)(const void *dev_buf, uint64_t nbytes, int dst);      /* like send, but the bytes live in DEVICE memory of this rank's GPU */
static struct { int rank, world; wtz_dist_bcast_fn bcast; wtz_dist_send_fn send; wtz_dist_recv_fn recv; wtz_dist_send_dev_fn send_dev; } g_dist = { 0, 1, NULL, NULL, NULL, NULL };
void wtzmo_set_dist(int rank, int world, wtz_dist_bcast_fn b, wtz_dist_send_fn sd, wtz_dist_recv_fn rv){
	g_dist.rank = rank; g_dist.world = world < 1 ? 1 : world; g_dist.bcast = b; g_dist.send = sd; g_dist.recv = rv; g_dist.send_dev = NULL;
}
/* optional: with this hook a rank > 0 hands its CIGAR text (the bulk of what travels: ~6.4 KB per record, 3.1 GB per configs[2] step) to the exchange straight
 * from the device buffer the library rendered it into (wtz_cigar_text_device) - device to device over xGMI on the nccl backend, no copy through this rank's host */
void wtzmo_set_dist_dev(wtz_dist_send_dev_fn f){ g_dist.send_dev = f; }
#define WTZ_DIST_MAX 16
enum { WTZ_CMD_DONE = 1, WTZ_CMD_PAIRS = 2, WTZ_CMD_CAND_BEGIN = 3, WTZ_CMD_CAND_END = 4, WTZ_CMD_GRP_BEGIN = 5, WTZ_CMD_GRP_END = 6,
       WTZ_CMD_ZIDX = 7,       /* the z-mer index of the batch: arg[0] queries (their ids follow as a broadcast) for the query-side index of every rank; arg[1] != 0: count[r] candidate
                                * reads (ids sent to rank r) for rank r's per-batch candidate-side index (--zindex-batch) */
       WTZ_CMD_ABORT = 8 };    /* rank 0 -> all: some rank failed; every rank leaves with exit code 1 (the reference's convention for every error: list.h:64-67) */
typedef struct { uint64_t cmd, arg[2], count[WTZ_DIST_MAX]; } wtz_dist_hdr_t;
/* Every reply of a rank > 0 starts with a status word: 0 = fine, 1 = scratch pool too small (the range is halved and redone), 2 = this rank failed (its own
 * stderr says why).  On 2, rank 0 finishes the round of the exchange it is in (the other ranks' replies are already on their way), broadcasts WTZ_CMD_ABORT
 * and every process exits 1 - nobody is left blocked in a receive. */
enum { WTZ_ST_OK = 0, WTZ_ST_AGAIN = 1, WTZ_ST_FAILED = 2 };

/* The pairs of a range are dealt to the PARTS of a batch, one part per GPU (--gpus N, or one rank per GPU; one part otherwise): pair
 * (q, c) belongs to part c % nparts.  Every device stage is pure in its pairs, so a part runs pair seeding, windows, alignment and
 * CIGAR rendering of its share on its own context, all parts side by side; the commit reads the results through PART_OF / LOCAL_OF
 * in the plan's order, i.e. the output does not depend on the number of parts.  Dealing by candidate (round 4; it was round-robin)
 * is what lets the z-mer index be split: part d walks only reads = d (mod nparts) as candidates, so its index holds the candidate side
 * of that residue class only (1 / nparts of the build) plus a small query-side index of the batch's queries (wtz_zindex_build_queries). */
typedef struct {
	wtz_ctx_t *ctx; int remote;                 /* remote > 0: the part is computed by that rank (ctx == NULL here) */
	uint8_t *xbuf; uint64_t xcap;              /* ranks: staging of one packed message (round 6: a request is ONE message, a reply a status word + ONE message + the CIGAR text) */
	uint32_t *pq, *pc; uint32_t npair, cappair;
	wtz_pair_summary_t *sum; uint64_t *box_off; wtz_winbox_t *boxes; uint64_t nbox, capbox;
	uint32_t *item_of; uint32_t *it_pair; uint8_t *it_dir; uint32_t nitem; wtz_aln_result_t *aln; char *cig; uint64_t ncig;
	void *cig_dev;               /* rank > 0 with a device-send hook: the text stays on the device (wtz_cigar_text_device) */
	int text_pending;           /* wtz_fetch_cigar_text_begin has been called for this range: part_text_wait() before the text is read */
	char *cigs[2]; uint64_t capcigs[2]; int cig_sel, cig_ext, ext_base;      /* two page-locked CIGAR text buffers, alternating per range; ext ids of the output writer */
	double t_call[6], t_io0;                   /* wall seconds of this part's device calls since the last fold into E (under E->mu) */
	struct eng_s *E; int again;                /* result of the last part_stages run (1 = scratch pool too small) */
	uint32_t *cq_ids, *cq_nr; uint64_t *cq_rows; uint32_t cq_n, cq_cap;      /* this device's share (every nparts-th query) of the candidate request in flight */
} part_t;
#define PIDX(part, local) (((uint32_t)(local) << 4) | (uint32_t)(part))      /* a pair of the plan: part (device / rank) in the low 4 bits, its index inside the part above */
#define PART_OF(b, g) (&(b)->parts[(g) & 15u])
#define CPART_OF(b, g) (&(b)->cparts[(g) & 15u])      /* commit side */
#define LOCAL_OF(b, g) ((g) >> 4)
#define DEAL_PART(np, q, c) ((c) % (np))               /* the part a pair is dealt to: by CANDIDATE id, so that a device only needs the candidate-side z-mers of its own residue class of reads */

typedef struct batch_s {       /* one batch in flight */
	eng_t *E; wtz_ctx_t *ctx; uint64_t seq;    /* ctx: the context of the candidate requests (= parts[0].ctx) */
	uint32_t *bq; uint32_t nbq, capbq;         /* queries in dispatch order */
	uint8_t *want;                              /* slot needs GPU work (not saturated when planned) */
	uint64_t *rows; uint32_t *nrow;             /* candidate heap arrays per slot (stride E->stride) */
	uint32_t *ids;
	part_t *parts; uint32_t nparts;
	part_t *cparts, *spare;                     /* cparts: the result arrays the commit reads (== parts unless ranges are pipelined: then the finished range's
	                                             * arrays are swapped into `spare` and the device stages of the next range fill `parts` meanwhile) */
	uint32_t npair, nitem;                      /* pairs / alignment items of the range over all parts */
	uint32_t *rowpair; size_t caprowpair;       /* pair index (in plan order) per (slot, row entry) */
	uint64_t spec_queries, used_queries;
	int holds_turn;
	/* candidates of the NEXT batch, requested before this batch is committed (single worker, no -G) */
	int pf_inflight; uint32_t *pf_ids; uint32_t pf_n, pf_cap; uint64_t *pf_rows; uint32_t *pf_nr; uint32_t pf_cursor_end;
	/* batches pipelined like ranges (single worker, one process): while the LAST range of this batch is committed, `alt` - a second batch on the same
	 * context(s) - has been formed from the prefetched candidates and runs its first range.  formed: batch_form() has run; started: its first range [0, start_s1) is
	 * already on the device (start_job) */
	struct batch_s *alt; int shares_ctx, formed, started; uint32_t start_s1; void *start_job; uint64_t masked_at_form;
} batch_t;

/* a device-stage failure ends the process at once: other host threads (index builders, parts) may still be inside HIP calls, and running the
 * exit handlers / the HIP runtime's teardown under them ends in a segmentation fault instead of exit code 1 */
#define DIE_NOW() do { fflush(NULL); _exit(1); } while(0)
#define DIE_WTZ(rc, what) do { if((rc) != WTZ_OK){ fprintf(stderr, " -- %s failed: %s --\n", what, wtz_last_error()); DIE_NOW(); } } while(0)

/* --repeat (benchmarking): the previous repeat's output file is removed by a side thread, see the repeat loop in wtzmo_main */
typedef struct { char *path; int pending, running; pthread_t th; } stale_job_t;
static void *stale_main(void *arg){ unlink((const char*)arg); return NULL; }
static void stale_start(stale_job_t *j){
	if(!j->pending || j->running) return;
	if(pthread_create(&j->th, NULL, stale_main, j->path) == 0) j->running = 1; else unlink(j->path);
	j->pending = 0;
}
static void stale_join(stale_job_t *j){ if(j->running){ pthread_join(j->th, NULL); j->running = 0; } }

static double now_s(void);
static uint32_t nbest_of(const eng_t *E, uint32_t id){
	uint32_t nb = (uint32_t)(((size_t)E->P.nbest) * E->rdlen[id] / E->avg_rdlen);      /* wtzmo.c:806-807 */
	return nb < E->P.nbest ? E->P.nbest : nb;
}

/* ---------------- output.  The commit only QUEUES a record (its integers + a pointer to the CIGAR text in the page-locked buffer that
 * wtz_fetch_cigar_text filled); a writer thread turns queued records into bytes - the 16 numeric columns as text (print_hits_wtzmo,
 * wtzmo.c:1235-1244) with the ~6 KB CIGAR text by reference in one writev(), or 64-byte binary records (--binary-out, include/wtz_ovlb.h) -
 * so that formatting (0.15-0.18 s of a 0.45 s commit at configs[2] when the commit thread did it) is off the one sequential stream of the run.
 * A part owns two text buffers and alternates between them per range; before a buffer is filled again it waits until every chunk that
 * points into it has been written (ext_busy). ---- */
#define OW_MAX_EXT 16
typedef struct {
	uint32_t pb1, pb2; int32_t tb, te, qb, qe, score, mat, mis, ins, del, aln;
	const char *cigar; uint32_t cigar_len; uint8_t dir2, kind; int8_t ext;      /* kind 0: an overlap record; 1: `cigar` is a heap string written verbatim (the "# ..." lines of -N) and freed; 2: a record whose CIGAR text is a heap copy, freed */
} orec_t;
typedef struct ochunk { orec_t *rec; int n, cap; size_t payload; unsigned ext_mask; uint64_t seq; struct ochunk *next; } ochunk_t;
#define OW_THREADS 3
typedef struct { pthread_t th; char *fmt; size_t capfmt; struct iovec *iov; size_t capiov; } owthread_t;      /* a writer thread's formatting buffer and vector */
/* OW_THREADS writer threads: each takes the next chunk, formats it (in parallel with the others), then waits for its TURN - the chunks' byte ranges follow
 * the queue order - and learns its file offset.  A regular file is written with pwritev() at that offset AFTER the turn is passed on, so the page-cache
 * copies of several chunks (3.4 GB per configs[2] step: 0.4 s on one thread, as long as a whole step would be on eight GPUs) run side by side; a pipe or a
 * terminal is written inside the turn, i.e. strictly in order. */
typedef struct {
	FILE *fp; int fd; pthread_mutex_t mu; pthread_cond_t cv, cv_ext, cv_turn;
	ochunk_t *head, *tail, *freelist, *cur; int done, started;
	int ext_busy[OW_MAX_EXT];
	const hx_read_t *reads; int binary;
	owthread_t T[OW_THREADS]; int nth;
	int seekable; uint64_t submit_seq, turn_seq; off_t off;          /* off: where the chunk whose turn it is starts */
	double t_format, t_write;                                        /* writer-thread seconds, summed over the threads (reported, not on the commit's clock) */
} owriter_t;
static owriter_t g_ow;
#define OCHUNK_RECS 8192
static int g_ochunk_recs = OCHUNK_RECS;      /* WTZ_OUT_CHUNK_RECS: test hook - tiny chunks put many of them in flight on the writer threads */
static int g_out_lag_ms = 0;                 /* WTZ_OUT_LAG_MS: test hook - a writer that lags behind the commit (a slow consumer), so that text buffers are refilled while records still point into them */
#define OCHUNK_PAYLOAD ((size_t)32 << 20)

static inline size_t put_str(char *o, const char *s){ size_t n = strlen(s); memcpy(o, s, n); return n; }
static inline size_t put_int(char *o, long long v){          /* as printf("%d") */
	char t[24]; int n = 0; size_t k = 0;
	unsigned long long u = v < 0 ? 0ULL - (unsigned long long)v : (unsigned long long)v;
	do { t[n++] = (char)('0' + u % 10); u /= 10; } while(u);
	if(v < 0) o[k++] = '-';
	while(n) o[k++] = t[--n];
	return k;
}
/* columns 1-16 + the tab in front of the CIGAR column (print_hits_wtzmo, wtzmo.c:1235-1244): the integer columns by hand (a 16-field sprintf per
 * record was half of the formatting time); the identity column stays with printf - its rounding of the binary quotient is part of the output */
static size_t format_record(const hx_read_t *reads, const orec_t *h, char *o){
	const int aln = h->aln == 0 ? 1 : h->aln;
	size_t k = 0;
	k += put_str(o + k, reads[h->pb1].name); o[k++] = '\t'; o[k++] = '+'; o[k++] = '\t';
	k += put_int(o + k, (long long)reads[h->pb1].len); o[k++] = '\t'; k += put_int(o + k, h->tb); o[k++] = '\t'; k += put_int(o + k, h->te); o[k++] = '\t';
	k += put_str(o + k, reads[h->pb2].name); o[k++] = '\t'; o[k++] = "+-"[h->dir2]; o[k++] = '\t';
	k += put_int(o + k, (long long)reads[h->pb2].len); o[k++] = '\t'; k += put_int(o + k, h->qb); o[k++] = '\t'; k += put_int(o + k, h->qe); o[k++] = '\t';
	k += put_int(o + k, h->score); o[k++] = '\t';
	k += (size_t)sprintf(o + k, "%0.3f", 1.0 * h->mat / aln); o[k++] = '\t';
	k += put_int(o + k, h->mat); o[k++] = '\t'; k += put_int(o + k, h->mis); o[k++] = '\t'; k += put_int(o + k, h->ins); o[k++] = '\t'; k += put_int(o + k, h->del); o[k++] = '\t';
	return k;
}
static void ow_write_all(owriter_t *w, struct iovec *iov, size_t niov, off_t at){      /* at < 0: at the descriptor's own position */
	for(size_t i = 0; i < niov;){
		size_t n = niov - i; if(n > 1024) n = 1024;
		ssize_t r = at < 0 ? writev(w->fd, iov + i, (int)n) : pwritev(w->fd, iov + i, (int)n, at);
		if(r > 0 && at >= 0) at += r;
		if(r < 0){ if(errno == EINTR) continue; fprintf(stderr, " -- write error on the output file: %s --\n", strerror(errno)); DIE_NOW(); }
		while(r > 0 && i < niov){        /* consume what was written; a partially written entry is advanced in place */
			if((size_t)r >= iov[i].iov_len){ r -= (ssize_t)iov[i].iov_len; i++; }
			else { iov[i].iov_base = (char*)iov[i].iov_base + r; iov[i].iov_len -= (size_t)r; r = 0; }
		}
		while(i < niov && iov[i].iov_len == 0) i++;
	}
}
typedef struct { owriter_t *w; int k; } owarg_t;
static void *owriter_main(void *arg){
	owriter_t *w = ((owarg_t*)arg)->w; owthread_t *me = &w->T[((owarg_t*)arg)->k];
	for(;;){
		pthread_mutex_lock(&w->mu);
		while(w->head == NULL && !w->done) pthread_cond_wait(&w->cv, &w->mu);
		ochunk_t *c = w->head;
		if(c == NULL){ pthread_mutex_unlock(&w->mu); break; }
		w->head = c->next; if(w->head == NULL) w->tail = NULL;
		pthread_mutex_unlock(&w->mu);
		if(g_out_lag_ms > 0){ struct timespec ts = { g_out_lag_ms / 1000, (long)(g_out_lag_ms % 1000) * 1000000L }; nanosleep(&ts, NULL); }
		const double tf0 = now_s();
		/* upper bound of the formatted bytes of the chunk (CIGAR text by reference does not count) */
		size_t need = 0;
		for(int i = 0; i < c->n; i++){ const orec_t *h = &c->rec[i]; need += w->binary ? sizeof(wtz_ovlb_rec_t) : (h->kind == 1 ? 0 : strlen(w->reads[h->pb1].name) + strlen(w->reads[h->pb2].name) + 256 + (h->cigar && h->cigar_len < 256 ? h->cigar_len : 0)); }
		if(need > me->capfmt){ me->capfmt = need + need / 2 + 4096; free(me->fmt); me->fmt = (char*)hx_realloc(NULL, me->capfmt); }
		if((size_t)c->n * 3 + 4 > me->capiov){ me->capiov = (size_t)c->n * 3 + 4; me->iov = (struct iovec*)hx_realloc(me->iov, sizeof(struct iovec) * me->capiov); }
		size_t k = 0, niov = 0, run0 = 0, total = 0;        /* [run0, k): formatted bytes not yet covered by a vector entry */
#define OW_FLUSH_RUN() do { if(k > run0){ me->iov[niov].iov_base = me->fmt + run0; me->iov[niov].iov_len = k - run0; niov++; run0 = k; } } while(0)
		for(int i = 0; i < c->n; i++){
			const orec_t *h = &c->rec[i];
			if(h->kind == 1){ OW_FLUSH_RUN(); if(!w->binary){ me->iov[niov].iov_base = (void*)h->cigar; me->iov[niov].iov_len = h->cigar_len; niov++; } continue; }
			if(w->binary){
				wtz_ovlb_rec_t r; memset(&r, 0, sizeof r);
				r.id1 = h->pb1; r.id2 = h->pb2; r.aln = (uint32_t)(h->aln == 0 ? 1 : h->aln); r.tb = h->tb; r.te = h->te; r.qb = h->qb; r.qe = h->qe; r.score = h->score;
				r.mat = h->mat; r.mis = h->mis; r.ins = h->ins; r.del = h->del; r.dir2 = h->dir2;
				memcpy(me->fmt + k, &r, sizeof r); k += sizeof r; continue;
			}
			k += format_record(w->reads, h, me->fmt + k);
			if(h->cigar == NULL){ me->fmt[k++] = '0'; me->fmt[k++] = 'M'; }
			else if(h->cigar_len < 256 && h->kind == 0){ memcpy(me->fmt + k, h->cigar, h->cigar_len); k += h->cigar_len; }
			else { OW_FLUSH_RUN(); me->iov[niov].iov_base = (void*)h->cigar; me->iov[niov].iov_len = h->cigar_len; niov++; }
			me->fmt[k++] = '\n';
		}
		OW_FLUSH_RUN();
#undef OW_FLUSH_RUN
		for(size_t i = 0; i < niov; i++) total += me->iov[i].iov_len;
		const double tf1 = now_s();
		/* the turn: chunks reach the stream in the order they were queued */
		pthread_mutex_lock(&w->mu);
		while(w->turn_seq != c->seq) pthread_cond_wait(&w->cv_turn, &w->mu);
		const off_t at = w->off;
		if(w->seekable){ w->off += (off_t)total; w->turn_seq++; pthread_cond_broadcast(&w->cv_turn); }
		pthread_mutex_unlock(&w->mu);
		const double tf2 = now_s();
		ow_write_all(w, me->iov, niov, w->seekable ? at : (off_t)-1);
		for(int i = 0; i < c->n; i++) if(c->rec[i].kind) free((void*)c->rec[i].cigar);
		pthread_mutex_lock(&w->mu);
		if(!w->seekable){ w->off += (off_t)total; w->turn_seq++; pthread_cond_broadcast(&w->cv_turn); }
		w->t_format += tf1 - tf0; w->t_write += now_s() - tf2;
		for(int e = 0; e < OW_MAX_EXT; e++) if(c->ext_mask & (1u << e)) w->ext_busy[e]--;
		if(c->ext_mask) pthread_cond_broadcast(&w->cv_ext);
		c->n = 0; c->payload = 0; c->ext_mask = 0; c->next = w->freelist; w->freelist = c;
		pthread_mutex_unlock(&w->mu);
	}
	return NULL;
}
static void out_start(FILE *fp, const hx_read_t *reads, int binary){
	owriter_t *w = &g_ow;
	static owarg_t args[OW_THREADS];
	if(!w->started){ pthread_mutex_init(&w->mu, NULL); pthread_cond_init(&w->cv, NULL); pthread_cond_init(&w->cv_ext, NULL); pthread_cond_init(&w->cv_turn, NULL); }
	fflush(fp);
	w->fp = fp; w->fd = fileno(fp); w->head = w->tail = NULL; w->cur = NULL; w->done = 0; w->started = 1; w->reads = reads; w->binary = binary; w->t_format = w->t_write = 0;
	memset(w->ext_busy, 0, sizeof w->ext_busy);
	{ struct stat sb; const off_t pos = lseek(w->fd, 0, SEEK_CUR);       /* a regular file (not opened for appending) takes positioned writes */
	  w->seekable = (pos >= 0 && fstat(w->fd, &sb) == 0 && S_ISREG(sb.st_mode) && !(fcntl(w->fd, F_GETFL) & O_APPEND) && !getenv("WTZ_OUT_SEQUENTIAL"));
	  w->off = w->seekable ? pos : 0; }
	w->submit_seq = w->turn_seq = 0;
	{ const char *e = getenv("WTZ_OUT_CHUNK_RECS"); if(e && atoi(e) > 0 && atoi(e) < OCHUNK_RECS) g_ochunk_recs = atoi(e); }
	{ const char *e = getenv("WTZ_OUT_LAG_MS"); g_out_lag_ms = (e && atoi(e) > 0) ? atoi(e) : 0; }
	w->nth = w->seekable ? OW_THREADS : 2;          /* a pipe still gets a second thread: formatting beside the write */
	for(int k = 0; k < w->nth; k++){ args[k].w = w; args[k].k = k; if(pthread_create(&w->T[k].th, NULL, owriter_main, &args[k]) != 0){ fprintf(stderr, " -- cannot start an output thread --\n"); DIE_NOW(); } }
}
static void out_submit(owriter_t *w){
	if(w->cur == NULL) return;
	pthread_mutex_lock(&w->mu);
	w->cur->next = NULL; w->cur->seq = w->submit_seq++;
	if(w->tail) w->tail->next = w->cur; else w->head = w->cur;
	w->tail = w->cur; w->cur = NULL;
	pthread_cond_signal(&w->cv);
	pthread_mutex_unlock(&w->mu);
}
/* the slot of the next queued record in the chunk under construction */
static orec_t *out_slot(size_t payload){
	owriter_t *w = &g_ow;
	if(w->cur && (w->cur->n >= g_ochunk_recs || w->cur->n == w->cur->cap || w->cur->payload > OCHUNK_PAYLOAD)) out_submit(w);
	if(w->cur == NULL){
		pthread_mutex_lock(&w->mu);
		ochunk_t *c = w->freelist; if(c) w->freelist = c->next;
		pthread_mutex_unlock(&w->mu);
		if(c == NULL){ c = (ochunk_t*)hx_realloc(NULL, sizeof(ochunk_t)); c->cap = OCHUNK_RECS; c->rec = (orec_t*)hx_realloc(NULL, sizeof(orec_t) * (size_t)c->cap); }
		c->n = 0; c->payload = 0; c->ext_mask = 0; c->next = NULL; w->cur = c;
	}
	w->cur->payload += payload;
	return &w->cur->rec[w->cur->n++];
}
/* a line written verbatim (takes ownership of the heap string) */
static void out_raw(char *text, size_t n){
	orec_t *r = out_slot(n); memset(r, 0, sizeof *r); r->kind = 1; r->cigar = text; r->cigar_len = (uint32_t)n; r->ext = -1;
}
/* the external buffer `ext` is about to be overwritten: everything that points into it must be on the stream.  The chunk under
 * construction (g_ow.cur) belongs to whoever holds E->mu (the committing worker) and is never looked at here: every commit hands
 * its chunk to the writer before it gives up E->mu (process_range), so all references into a worker's buffer are already queued
 * when that worker comes back to refill it. */
static void out_wait_ext(int ext){
	owriter_t *w = &g_ow;
	if(!w->started || ext < 0 || ext >= OW_MAX_EXT) return;
	pthread_mutex_lock(&w->mu);
	while(w->ext_busy[ext] > 0) pthread_cond_wait(&w->cv_ext, &w->mu);
	pthread_mutex_unlock(&w->mu);
}
/* everything queued so far is on the stream when this returns */
static void out_finish(void){
	owriter_t *w = &g_ow;
	out_submit(w);
	pthread_mutex_lock(&w->mu); w->done = 1; pthread_cond_broadcast(&w->cv); pthread_mutex_unlock(&w->mu);
	for(int k = 0; k < w->nth; k++) pthread_join(w->T[k].th, NULL);
	if(w->seekable && lseek(w->fd, w->off, SEEK_SET) < 0){ fprintf(stderr, " -- cannot position the output file: %s --\n", strerror(errno)); DIE_NOW(); }      /* whatever follows through the FILE continues behind the records */
}

/* ---------------- record writer + state merge (wtzmo.c:1170-1249, 1319-1329) ---------------- */
static void order_push(eng_t *E, uint64_t v){
	if(!E->keep_order) return;
	if(E->n_order == E->cap_order){ E->cap_order = E->cap_order ? E->cap_order * 2 : 4096; E->closed_order = (uint64_t*)hx_realloc(E->closed_order, 8 * E->cap_order); }
	E->closed_order[E->n_order++] = v;
}
static void flush_pending(eng_t *E){
	pending_t *p = &E->pend;
	const hx_read_t *reads = E->st.reads;
	if(p->rd_id != 0xFFFFFFFFu){
		if(!E->do_align){
			for(size_t i = 0; i < p->nseed; i++){
				const seed_t *s = &p->seeds[i];
				if(s->closed) continue;
				char *o = (char*)hx_realloc(NULL, strlen(reads[p->rd_id].name) + strlen(reads[s->pb2].name) + 96);
				out_raw(o, (size_t)sprintf(o, "# %s\t%c\t%d\t%s\t%c\t%d\t%d\n", reads[p->rd_id].name, '+', reads[p->rd_id].len,
					reads[s->pb2].name, "+-"[s->dir], reads[s->pb2].len, s->ovl));
			}
		} else {
			for(size_t j = 0; j < p->nhit; j++){
				hit_t *h = &p->hits[j];
				E->nrec++;
				if(h->aln == 0) h->aln = 1;
				uint32_t x1 = (uint32_t)(h->tb < h->qb ? h->tb : h->qb);
				int r1 = (int)reads[h->pb1].len - h->te, r2 = (int)reads[h->pb2].len - h->qe;
				uint32_t x2 = (uint32_t)(r1 < r2 ? r1 : r2);
				if(x1 + x2 <= 200u){ E->rdcovs[h->pb1]++; E->rdcovs[h->pb2]++; }          /* max_unalign_in_dovetail, wtzmo.c:175 */
				/* the record itself was queued for the writer thread when the hit was committed (emit_record): same order, same bytes */
			}
		}
	}
	p->nhit = 0; p->nseed = 0;
	for(size_t i = 0; i < p->nmask; i++){ if(!E->masked[p->masks[i]]) E->n_masked++; E->masked[p->masks[i]] = 1; }
	p->nmask = 0;
	for(size_t i = 0; i < p->nclosed; i++){
		if(hx_set_put(&E->closed, p->closed[i])){
			order_push(E, p->closed[i]);
			uint32_t a = (uint32_t)(p->closed[i] >> 33), b = (uint32_t)((p->closed[i] & 0xFFFFFFFFu) >> 1);
			E->pair_bp += (uint64_t)E->rdlen[a] + E->rdlen[b]; E->n_pairs++;
			/* behind the insertion: a helper that sees the new count sees the pair (cq_spec_*) */
			__atomic_store_n(&E->closed_touch[a], E->closed_touch[a] + 1u, __ATOMIC_RELEASE); __atomic_store_n(&E->closed_touch[b], E->closed_touch[b] + 1u, __ATOMIC_RELEASE);      /* one writer: a release store, not a locked read-modify-write */
		}
	}
	p->nclosed = 0;
}

static void pend_mask(pending_t *p, uint32_t id){
	for(size_t i = 0; i < p->nmask; i++) if(p->masks[i] == id) return;
	if(p->nmask == p->capmask){ p->capmask = p->capmask ? p->capmask * 2 : 16; p->masks = (uint32_t*)hx_realloc(p->masks, p->capmask * 4); }
	p->masks[p->nmask++] = id;
}
/* what flush_pending will touch for this pair, requested now (one query early): its slot of the closed-pair table (a random line of a table of tens of MB: the merge of
 * a configs[2] step's 487 000 pairs was 0.07 s of misses) and the two reads' counters */
static eng_t *g_pend_eng = NULL;
static inline void pend_prefetch_pair(uint64_t v){
	const eng_t *E = g_pend_eng;
	if(!E || !E->closed.cap) return;
	__builtin_prefetch(&E->closed.tab[hx_mix(v) & (E->closed.cap - 1)], 1, 1);
	const uint32_t a = (uint32_t)(v >> 33), b = (uint32_t)((v & 0xFFFFFFFFu) >> 1);
	__builtin_prefetch(&E->closed_touch[a], 1, 1); __builtin_prefetch(&E->closed_touch[b], 1, 1);
}
static void pend_closed(pending_t *p, uint64_t v){
	pend_prefetch_pair(v);
	if(p->nclosed == p->capclosed){ p->capclosed = p->capclosed ? p->capclosed * 2 : 64; p->closed = (uint64_t*)hx_realloc(p->closed, p->capclosed * 8); }
	p->closed[p->nclosed++] = v;
}
static void pend_hit(pending_t *p, const hit_t *h){
	if(g_pend_eng){ __builtin_prefetch(&g_pend_eng->rdcovs[h->pb2], 1, 1); }
	if(p->nhit == p->caphit){ p->caphit = p->caphit ? p->caphit * 2 : 64; p->hits = (hit_t*)hx_realloc(p->hits, p->caphit * sizeof(hit_t)); }
	p->hits[p->nhit++] = *h;
}

/* one .ovl line (print_hits_wtzmo, wtzmo.c:1170-1249) queued for the writer thread; cigar == NULL prints "0M" */
static void emit_record(eng_t *E, const hit_t *h, const char *cigar, size_t cigar_len, int ext){
	(void)E;
	orec_t *r = out_slot(cigar_len + 128);
	r->pb1 = h->pb1; r->pb2 = h->pb2; r->tb = h->tb; r->te = h->te; r->qb = h->qb; r->qe = h->qe; r->score = h->score;
	r->mat = h->mat; r->mis = h->mis; r->ins = h->ins; r->del = h->del; r->aln = h->aln; r->dir2 = (uint8_t)h->dir2; r->kind = 0;
	r->cigar = cigar; r->cigar_len = (uint32_t)cigar_len; r->ext = (int8_t)ext;
	/* The text is read LATER, on a writer thread - the short ones too (they are copied into the formatted run there) - so every record that points into a
	 * text buffer holds that buffer (ext_busy) until its chunk is on the stream, whatever its length: a chunk of short records only used to leave the
	 * buffer unprotected, and the part refilled (or freed) it two ranges later under a writer that lagged behind a slow consumer. */
	if(cigar && cigar_len > 0 && !(ext >= 0 && ext < OW_MAX_EXT)){      /* a text buffer the writer does not track (parts beyond OW_MAX_EXT / 2): the record takes a copy */
		char *cp = (char*)hx_realloc(NULL, cigar_len); memcpy(cp, cigar, cigar_len); r->cigar = cp; r->kind = 2;
	} else if(cigar && cigar_len > 0){
		owriter_t *w = &g_ow; ochunk_t *c = w->cur;
		if(!(c->ext_mask & (1u << ext))){ c->ext_mask |= 1u << ext; pthread_mutex_lock(&w->mu); w->ext_busy[ext]++; pthread_mutex_unlock(&w->mu); }
	}
}

__attribute__((unused)) static char *cigar_text(const uint32_t *c, uint32_t n){        /* kswx.h:1093-1120 */
	size_t cap = (size_t)n * 11 + 2, k = 0; char *s = (char*)hx_realloc(NULL, cap);
	for(uint32_t i = 0; i < n; i++){
		uint32_t op = c[i] & 0xF, len = c[i] >> 4;
		if(len == 0) continue;
		if(op > 2){ fprintf(stderr, " -- CIGAR only support M(0),I(1),D(2) cigar, but met ?(%u) --\n", op); DIE_NOW(); }
		char d[12]; int nd = 0;
		while(len){ d[nd++] = (char)('0' + len % 10); len /= 10; }
		while(nd) s[k++] = d[--nd];
		s[k++] = "MID"[op];
	}
	s[k] = 0; return s;
}

/* a[i] <- a[lo] + ... + a[i] for i in [lo, hi), 16-bit wrap-around; lo is a multiple of 8 */
#ifdef __SSE2__
#include <emmintrin.h>
#endif
static void depth_prefix_u16(uint16_t *a, size_t lo, size_t hi){
	size_t i = lo; uint16_t run = 0;
#ifdef __SSE2__
	__m128i carry = _mm_setzero_si128();
	for(; i + 8 <= hi; i += 8){
		__m128i x = _mm_loadu_si128((const __m128i*)(a + i));
		x = _mm_add_epi16(x, _mm_slli_si128(x, 2)); x = _mm_add_epi16(x, _mm_slli_si128(x, 4)); x = _mm_add_epi16(x, _mm_slli_si128(x, 8));
		x = _mm_add_epi16(x, carry);
		_mm_storeu_si128((__m128i*)(a + i), x);
		carry = _mm_shuffle_epi32(_mm_shufflehi_epi16(x, 0xFF), 0xFF);      /* the last sum in every word */
	}
	if(i > lo) run = a[i - 1];
#endif
	for(; i < hi; i++){ run = (uint16_t)(run + a[i]); a[i] = run; }
}

/* weights[i] of wtzmo.c:933-936: float = (depth rule); then float = float * (double positional factor) */
static inline float rep_weight(const uint16_t *windeps, const wtz_params_c *P, int alen, int i){
	float w = (windeps[i] <= P->win_rep_norm) ? 1.0 : ((windeps[i] >= P->win_rep_cutoff) ? 0.0 : P->win_rep_norm / (float)windeps[i]);
	int df = i < alen / 2 ? alen / 2 - i : i - alen / 2;
	w = w * (0.3 + 0.7 * (df / (alen / 2.0)));
	return w;
}

/* The commit of a query walks its pairs' summaries and window boxes - results that arrived from the device a moment ago and are in no cache - in candidate order:
 * 72 of the commit's 240 ms per configs[2] step were those misses (round 6, timers in parts).  The lines of the NEXT query's pairs are requested while this one is committed. */
static void commit_prefetch(const eng_t *E, const batch_t *b, uint32_t slot){
	if(!b->want[slot] || E->masked[b->bq[slot]]) return;
	const uint32_t n = b->nrow[slot]; const int dm = E->P.dot_matrix;
	for(uint32_t k = 0; k < n; k++){
		const uint32_t g = b->rowpair[(size_t)slot * E->stride + k];
		if(g == 0xFFFFFFFFu) continue;
		const part_t *pt = CPART_OF(b, g); const uint32_t li = LOCAL_OF(b, g);
		__builtin_prefetch(&pt->sum[li], 0, 1);
		if(!dm){ __builtin_prefetch(&pt->box_off[(size_t)li * 2], 0, 1); if(pt->item_of) __builtin_prefetch(&pt->item_of[li], 0, 1); }
	}
}
/* ---------------- commit of one query over the batch results (wtzmo.c:806-1130) ---------------- */
/* The part that does not depend on the ORDER of the commits, only on the set of closed pairs: candidates filtered by that set, in the reference's order, trimmed
 * (wtzmo.c:813-822); window depth, repeat-weighted seeds, seed order (wtzmo.c:905-986).  `racy`: called by a helper thread beside the committing thread (cq_spec_*). */
static void cq_prepare(const eng_t *E, const batch_t *b, uint32_t slot, cq_work_t *w, int racy){
	const wtz_params_c *P = &E->P;
	const uint32_t pbid = b->bq[slot];
	const int alen = (int)E->rdlen[pbid];
	const double tq0 = now_s();
	uint32_t nc = b->nrow[slot];
	if(w->capcand < (size_t)nc + 1){ w->capcand = ((size_t)nc + 1) * 2; w->cand = (cand_t*)hx_realloc(w->cand, sizeof(cand_t) * w->capcand); }
	cand_t *cand = w->cand;
	for(uint32_t i = 0; i < nc; i++){
		cand[i].e = b->rows[(size_t)slot * E->stride + i]; cand[i].pidx = b->rowpair[(size_t)slot * E->stride + i]; cand[i].pad = 0;
		const uint64_t key = hx_pair_key(pbid, cand[i].e >> 32);
		if(racy ? hx_set_has_racy(&E->closed, key) : hx_set_has(&E->closed, key)) cand[i].e &= 0xFFFFFFFF00000000ULL;
	}
	sort_cands_exact(cand, nc);
	while(nc && (uint32_t)cand[nc - 1].e == 0) nc--;
	w->nc = nc; w->nseed = 0; w->ngate = 0;
	const double tq1 = now_s(); w->t[0] = tq1 - tq0; w->t[1] = w->t[2] = w->t[3] = w->t[4] = 0;
	if(P->dot_matrix) return;
	for(uint32_t i = 0; i < nc; i++){       /* the window boxes of the pairs that are walked below (their summaries were requested one query ago: commit_prefetch) */
		if(cand[i].pidx == 0xFFFFFFFFu) continue;
		const part_t *pt = CPART_OF(b, cand[i].pidx); const uint32_t li = LOCAL_OF(b, cand[i].pidx);
		const wtz_pair_summary_t *S = &pt->sum[li];
		if(!S->gate) continue;
		const wtz_winbox_t *bx = pt->boxes + pt->box_off[(size_t)li * 2];
		const uint32_t nb = S->nwin[0] + S->nwin[1];
		for(uint32_t k = 0; k < nb; k += 4) __builtin_prefetch(bx + k, 0, 1);      /* 16-byte boxes: four per line */
	}
	if(w->capdep < (size_t)alen + 24){      /* zero from the start and after every query */
		w->capdep = ((size_t)alen + 24) * 2; free(w->dep);
		w->dep = (uint16_t*)calloc(w->capdep, 2); if(!w->dep){ fprintf(stderr, " -- Out of memory --\n"); DIE_NOW(); }
	}
	uint16_t *windeps = w->dep;
	int dep_lo = alen, dep_hi = 0;             /* span of the marks */
	if(w->capseeds < (size_t)nc + 1){ w->capseeds = ((size_t)nc + 1) * 2; w->seeds = (seed_t*)hx_realloc(w->seeds, sizeof(seed_t) * w->capseeds); }
	seed_t *seeds = w->seeds; uint32_t nseed = 0, ngate = 0;
	for(uint32_t i = 0; i < nc; i++){
		const uint32_t id2 = (uint32_t)(cand[i].e >> 32);
		if(cand[i].pidx == 0xFFFFFFFFu){ fprintf(stderr, " -- internal error: pair (%u,%u) missing from the batch plan --\n", pbid, id2); DIE_NOW(); }
		const part_t *pt = CPART_OF(b, cand[i].pidx); const uint32_t li = LOCAL_OF(b, cand[i].pidx);
		const wtz_pair_summary_t *S = &pt->sum[li];
		if(!S->gate) continue;
		ngate++;
		for(uint32_t dir = 0; dir < 2; dir++){
			const wtz_winbox_t *bx = pt->boxes + pt->box_off[(size_t)li * 2 + dir];
			for(uint32_t k = 0; k < S->nwin[dir]; k++){       /* wtzmo.c:908 increments windeps over [beg,end): kept as +1/-1 marks, summed below */
				const int wb = bx[k].beg[0], we = bx[k].end[0];
				if(wb < we){ windeps[wb]++; windeps[we]--; if(wb < dep_lo) dep_lo = wb; if(we > dep_hi) dep_hi = we; }
			}
		}
		const uint32_t dir = (S->ovl[0] < S->ovl[1]);
		if(S->ovl[dir] >= P->ztot){ seed_t sd; sd.pb2 = id2; sd.dir = dir; sd.ovl = S->ovl[dir]; sd.closed = 0; sd.pidx = cand[i].pidx; seeds[nseed++] = sd; }
	}
	const double tqa = now_s(); w->t[1] = tqa - tq1;
	/* running depth over the span of the marks (it is zero outside: every interval is closed); arithmetic modulo 2^16 like the reference's u2i counters */
	if(dep_lo < dep_hi) depth_prefix_u16(windeps, (size_t)(dep_lo & ~7), (size_t)dep_hi + 1);
	const double tqb = now_s(); w->t[2] = tqb - tqa;
	/* repeat weighting: the reference fills weights[0..alen) (wtzmo.c:933-936) but only reads the entry at the middle of each
	 * window (954): evaluated on demand by rep_weight() with the same float/double mix */
	for(uint32_t i = 0; i < nseed; i++){
		seed_t *sd = &seeds[i];
		const int blen = (int)E->rdlen[sd->pb2];
		const part_t *pt = CPART_OF(b, sd->pidx); const uint32_t li = LOCAL_OF(b, sd->pidx);
		const wtz_pair_summary_t *S = &pt->sum[li];
		const wtz_winbox_t *bx = pt->boxes + pt->box_off[(size_t)li * 2 + sd->dir];
		uint32_t ol = 0; double avg;
		for(uint32_t k = 0; k < S->nwin[sd->dir]; k++){
			avg = (bx[k].end[0] - bx[k].beg[0]) * rep_weight(windeps, P, alen, (bx[k].beg[0] + bx[k].end[0]) / 2);
			int mid = (int)((bx[k].beg[1] + bx[k].end[1]) / 2);
			int df = mid < blen / 2 ? blen / 2 - mid : mid - blen / 2;
			avg = avg * (0.3 + 0.7 * (df / (blen / 2.0)));
			ol += avg;
		}
		sd->ovl = ol & 0x1FFFFFFFu;
		if(ol * P->win_rep_cutoff < P->ztot * P->win_rep_norm) sd->closed = 1;                 /* wtzmo.c:964 */
	}
	const double tqc = now_s(); w->t[3] = tqc - tqb;
	sort_seeds_exact(seeds, nseed);
	if(dep_lo < dep_hi) memset(windeps + (dep_lo & ~7), 0, 2 * (size_t)(dep_hi + 1 - (dep_lo & ~7)));
	w->nseed = nseed; w->ngate = ngate; w->t[4] = now_s() - tqc;
}

/* ---- queries prepared ahead of the commit by helper threads (round 6).  The prepared part of a query depends on the closed-pair set only through the pairs that
 * hold the query's own read, so: every pair that enters the set bumps a counter of both its reads (flush_pending, AFTER the insertion); a helper notes the
 * query's counter BEFORE it reads the set; the committing thread takes a prepared query iff the counter has not moved since - otherwise it prepares the query
 * again itself, as before.  The set is read while the committing thread inserts into it: no growth happens meanwhile (hx_set_reserve in front of the range),
 * and a slot goes from empty to a key in one aligned store.  WTZ_COMMIT_HELPERS=<n> (default 4, 0 = none). ---- */
#define CQ_RING 64u
typedef struct { cq_work_t w; uint32_t v0; int skipped; uint64_t ready_for; } cq_ent_t;      /* ready_for == slot + 1: filled for that slot */
typedef struct cq_spec_s {
	eng_t *E; const batch_t *b; uint32_t s0, s1;
	uint64_t next, done; int stop, active, nth;        /* next: the slot a helper takes next; done: slots below it are consumed (a ring entry is free again) */
	pthread_t th[16]; cq_ent_t ring[CQ_RING];
} cq_spec_t;
static void *cq_spec_main(void *arg){
	cq_spec_t *sp = (cq_spec_t*)arg; eng_t *E = sp->E; const batch_t *b = sp->b;
	for(;;){
		uint64_t s = __atomic_load_n(&sp->next, __ATOMIC_RELAXED);
		if(s >= sp->s1 || __atomic_load_n(&sp->stop, __ATOMIC_RELAXED)) break;
		if(s >= __atomic_load_n(&sp->done, __ATOMIC_ACQUIRE) + CQ_RING){ __builtin_ia32_pause(); continue; }      /* the ring is full: the commit has to catch up */
		if(!__atomic_compare_exchange_n(&sp->next, &s, s + 1, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) continue;
		cq_ent_t *en = &sp->ring[s % CQ_RING];
		/* the entry's previous occupant (slot s - CQ_RING) may have been passed by the commit without being waited for (a masked or saturated query) while its helper is
		 * still at work: it has been taken before this slot and will finish */
		if(s >= (uint64_t)sp->s0 + CQ_RING) while(__atomic_load_n(&en->ready_for, __ATOMIC_ACQUIRE) != s - CQ_RING + 1) __builtin_ia32_pause();
		const uint32_t pbid = b->bq[s];
		en->skipped = (!b->want[s] || __atomic_load_n(&E->masked[pbid], __ATOMIC_RELAXED)) ? 1 : 0;
		if(!en->skipped){
			en->v0 = __atomic_load_n(&E->closed_touch[pbid], __ATOMIC_ACQUIRE);
			cq_prepare(E, b, (uint32_t)s, &en->w, 1);
		}
		__atomic_store_n(&en->ready_for, s + 1, __ATOMIC_RELEASE);
	}
	return NULL;
}
static void cq_spec_start(eng_t *E, const batch_t *b, uint32_t s0, uint32_t s1, uint64_t npair){
	static int nth = -1;
	if(nth < 0){ const char *e = getenv("WTZ_COMMIT_HELPERS"); nth = e ? atoi(e) : 4; if(nth < 0) nth = 0; if(nth > 16) nth = 16; }
	if(!E->spec){ E->spec = (cq_spec_t*)calloc(1, sizeof(cq_spec_t)); if(!E->spec) DIE_NOW(); }
	cq_spec_t *sp = E->spec;
	sp->active = 0;
	if(nth == 0 || E->rows_all || s1 < s0 + 2) return;
	hx_set_reserve(&E->closed, (size_t)npair * 2 + 4096);      /* every pair this commit can close, with room: the table must not grow under the helpers */
	sp->E = E; sp->b = b; sp->s0 = s0; sp->s1 = s1; sp->next = s0; sp->done = s0; sp->stop = 0; sp->nth = 0;
	for(uint32_t k = 0; k < CQ_RING; k++) sp->ring[k].ready_for = 0;
	for(int k = 0; k < nth; k++){ if(pthread_create(&sp->th[sp->nth], NULL, cq_spec_main, sp) == 0) sp->nth++; }
	sp->active = sp->nth > 0;
}
static inline void cq_spec_done(eng_t *E, uint32_t slot){ if(E->spec && E->spec->active) __atomic_store_n(&E->spec->done, (uint64_t)slot + 1, __ATOMIC_RELEASE); }
static void cq_spec_stop(eng_t *E){
	cq_spec_t *sp = E->spec;
	if(!sp || !sp->active) return;
	__atomic_store_n(&sp->stop, 1, __ATOMIC_RELAXED);
	for(int k = 0; k < sp->nth; k++) pthread_join(sp->th[k], NULL);
	sp->active = 0;
}
/* the prepared part of query `slot`: a helper's if it is still valid, else made here */
static cq_work_t *cq_take(eng_t *E, batch_t *b, uint32_t slot){
	cq_spec_t *sp = E->spec;
	if(sp && sp->active){
		cq_ent_t *en = &sp->ring[slot % CQ_RING];
		const double t0 = now_s();
		while(__atomic_load_n(&en->ready_for, __ATOMIC_ACQUIRE) != (uint64_t)slot + 1) __builtin_ia32_pause();      /* slots are taken in order: some helper has this one */
		E->t_spec_wait += now_s() - t0;
		if(!en->skipped && __atomic_load_n(&E->closed_touch[b->bq[slot]], __ATOMIC_ACQUIRE) == en->v0){ E->n_spec_hit++; return &en->w; }
		E->n_spec_miss++;
	}
	cq_prepare(E, b, slot, &E->cq, 0);
	E->t_cq[0] += E->cq.t[0]; E->t_cq[1] += E->cq.t[1] + E->cq.t[2] + E->cq.t[3] + E->cq.t[4];
	for(int k = 0; k < 4; k++) E->t_cq1[k] += E->cq.t[k + 1];
	return &E->cq;
}

static void commit_query(eng_t *E, batch_t *b, uint32_t slot){
	const wtz_params_c *P = &E->P;
	pending_t *pd = &E->pend;
	const uint32_t pbid = b->bq[slot];
	pd->rd_id = pbid;
	const uint32_t nbest = nbest_of(E, pbid);
	uint32_t bcov = E->rdcovs[pbid];
	if(bcov >= nbest) return;
	E->used_queries++; b->used_queries++;
	if(!b->want[slot]){ fprintf(stderr, " -- internal error: no candidate row for read %u --\n", pbid); DIE_NOW(); }
	cq_work_t *w = cq_take(E, b, slot);
	const double tq2 = now_s();
	cand_t *cand = w->cand; const uint32_t nc = w->nc;
	if(E->rows_all){       /* -G: the trimmed, sorted list persists as the reference's rdhits entry */
		for(uint32_t i = 0; i < nc; i++) E->rows[(size_t)pbid * E->stride + i] = cand[i].e;
		E->nrow[pbid] = nc;
	}
	if(P->dot_matrix){
		for(uint32_t i = 0; i < nc; i++){
			const uint32_t id2 = (uint32_t)(cand[i].e >> 32);
			if(cand[i].pidx == 0xFFFFFFFFu){ fprintf(stderr, " -- internal error: pair (%u,%u) missing from the batch plan --\n", pbid, id2); DIE_NOW(); }
			const wtz_pair_summary_t *S = &CPART_OF(b, cand[i].pidx)->sum[LOCAL_OF(b, cand[i].pidx)];
			if(!S->gate) continue;
			E->used_pairs++;
			pend_closed(pd, hx_pair_key(id2, pbid));
			int d1 = S->dm_qe - S->dm_qb, d2 = S->dm_te - S->dm_tb;
			uint32_t ol = (uint32_t)(d1 > d2 ? d1 : d2);
			/* -N: print_hits_wtzmo gets the (empty) seed list and drops the dot-matrix hits unprinted and uncounted (wtzmo.c:1175-1210, 1319) */
			if(E->do_align && S->dm_score >= P->min_score && S->dm_score >= (int)(P->min_id * ol)){
				hit_t H; memset(&H, 0, sizeof H);
				H.pb1 = pbid; H.pb2 = id2; H.dir2 = (uint32_t)S->dm_dir; H.score = S->dm_score;
				H.tb = S->dm_tb; H.te = S->dm_te; H.qb = S->dm_qb; H.qe = S->dm_qe; H.mat = S->dm_score; H.aln = (int)ol;
				pend_hit(pd, &H);
				emit_record(E, &H, NULL, 0, -1);
			}
		}
		E->t_cq[2] += now_s() - tq2;
		return;
	}
	E->used_pairs += w->ngate;
	seed_t *seeds = w->seeds; const uint32_t nseed = w->nseed;
	if(!E->do_align){
		if(pd->capseed < nseed){ pd->capseed = nseed; pd->seeds = (seed_t*)hx_realloc(pd->seeds, sizeof(seed_t) * nseed); }
		memcpy(pd->seeds, seeds, sizeof(seed_t) * nseed); pd->nseed = nseed;
	} else {
		uint32_t ncand = P->ncand;
		for(uint32_t i = 0; i < nseed && i < ncand; i++){      /* the alignment results the loop below reads */
			const part_t *pt = CPART_OF(b, seeds[i].pidx); const uint32_t item = pt->item_of[LOCAL_OF(b, seeds[i].pidx)];
			if(item != 0xFFFFFFFFu) __builtin_prefetch(&pt->aln[item], 0, 1);
		}
		for(uint32_t i = 0; i < nseed && i < ncand; i++){
			seed_t *s = &seeds[i];
			if(s->closed){ ncand++; continue; }
			pend_closed(pd, hx_pair_key(s->pb2, pbid));
			const part_t *pt = CPART_OF(b, s->pidx);
			const uint32_t item = pt->item_of[LOCAL_OF(b, s->pidx)];
			if(item == 0xFFFFFFFFu || pt->it_dir[item] != s->dir){ fprintf(stderr, " -- internal error: alignment of (%u,%u) missing from the batch plan --\n", pbid, s->pb2); DIE_NOW(); }
			E->used_items++;
			const wtz_aln_result_t *x = &pt->aln[item];
			if(x->n_regs == 0){ s->closed = 1; ncand++; continue; }
			if(x->score < P->min_score || x->mat < x->aln * P->min_id) continue;
			hit_t H; memset(&H, 0, sizeof H);
			H.pb1 = pbid; H.pb2 = s->pb2; H.dir2 = s->dir; H.score = x->score;
			H.tb = x->tb; H.te = x->te; H.qb = x->qb; H.qe = x->qe; H.mat = x->mat; H.mis = x->mis; H.ins = x->ins; H.del = x->del; H.aln = x->aln;
			pend_hit(pd, &H);
			if(E->binary_out) emit_record(E, &H, NULL, 0, -1); else emit_record(E, &H, pt->cig + x->text_off, x->text_len, pt->cig_ext);
			{   /* dovetail / containment bookkeeping (wtzmo.c:1065-1100) */
				const uint32_t len1 = E->rdlen[H.pb1], len2 = E->rdlen[H.pb2];
				uint32_t x1 = (uint32_t)(H.tb < H.qb ? H.tb : H.qb);
				int r1 = (int)len1 - H.te, r2 = (int)len2 - H.qe;
				uint32_t x2 = (uint32_t)(r1 < r2 ? r1 : r2);
				if(x1 + x2 <= 200u){
					uint32_t x3 = ((H.tb == 0 && H.qb) || (H.te == (int)len1 && H.qe < (int)len2));
					uint32_t x4 = ((H.qb == 0 && H.tb) || (H.qe == (int)len2 && H.te < (int)len1));
					x1 = len2 + (uint32_t)H.qb - (uint32_t)H.qe;
					x2 = len1 + (uint32_t)H.tb - (uint32_t)H.te;
					if(x1 <= 0u && x3 == 0){               /* max_unalign_in_contained = 0, wtzmo.c:174 */
						if(x2 <= 0u && x4 == 0){
							if(len1 > len2){ pend_mask(pd, H.pb2); }
							else if(len1 < len2){ pend_mask(pd, H.pb1); break; }
							else if(H.pb2 > H.pb1){ pend_mask(pd, H.pb2); continue; }
							else { pend_mask(pd, H.pb1); break; }
						} else { pend_mask(pd, H.pb2); continue; }
						ncand++;
					} else if(x2 <= 0u && x4 == 0){ pend_mask(pd, H.pb1); break; }
					bcov++;
					if(bcov >= nbest) break;
				}
			}
		}
	}
	E->t_cq[2] += now_s() - tq2;
}

/* ---------------- one batch: plan -> GPU -> commit ---------------- */
/* a device stage of a part failed: 1 = scratch pool too small (nothing changed, the range is halved), otherwise fatal - at once in a one-process run; between
 * ranks the part reports WTZ_ST_FAILED so that the exchange in flight is finished and rank 0 can end every rank in-band (ranks_abort) */
#define TRY_WTZ(rc, what) do { if((rc) == WTZ_E_POOL) return WTZ_ST_AGAIN; \
	if((rc) != WTZ_OK && g_dist.world > 1){ fprintf(stderr, " -- rank %d: %s failed: %s --\n", g_dist.rank, what, wtz_last_error()); return WTZ_ST_FAILED; } DIE_WTZ(rc, what); } while(0)

/* under E->mu: pairs of slots [s0,s1) whose candidate pair is not closed right now */
static void plan_pairs(eng_t *E, batch_t *b, uint32_t s0, uint32_t s1){
	const double tp0 = now_s();
	b->npair = 0;
	for(uint32_t d = 0; d < b->nparts; d++) b->parts[d].npair = 0;
	if((size_t)b->nbq * E->stride > b->caprowpair){ b->caprowpair = (size_t)b->nbq * E->stride; b->rowpair = (uint32_t*)hx_realloc(b->rowpair, b->caprowpair * 4); }
	uint64_t rows_seen = 0;
	for(uint32_t s = s0; s < s1; s++){
		const uint32_t q = b->bq[s];
		rows_seen += b->want[s] ? b->nrow[s] : 0;              /* what range_end counted */
		if(!b->want[s] || E->masked[q] || E->rdcovs[q] >= nbest_of(E, q)) continue;
		for(uint32_t k = 0; k < b->nrow[s]; k++){
			const uint64_t e = b->rows[(size_t)s * E->stride + k];
			const uint32_t id2 = (uint32_t)(e >> 32);
			b->rowpair[(size_t)s * E->stride + k] = 0xFFFFFFFFu;
			if((uint32_t)e == 0 || id2 == 0xFFFFFFFFu) continue;
			if(hx_set_has(&E->closed, hx_pair_key(q, id2))) continue;
			const uint32_t dpart = DEAL_PART(b->nparts, q, id2);
			part_t *pt = &b->parts[dpart];
			if(pt->npair == pt->cappair){ pt->cappair = pt->cappair ? pt->cappair * 2 : 4096; pt->pq = (uint32_t*)hx_realloc(pt->pq, pt->cappair * 4); pt->pc = (uint32_t*)hx_realloc(pt->pc, pt->cappair * 4); }
			if(pt->npair >= (1u << 28)){ fprintf(stderr, " -- more than 2^28 pairs of one range on one device --\n"); DIE_NOW(); }
			pt->pq[pt->npair] = q; pt->pc[pt->npair] = id2; pt->npair++;
			b->rowpair[(size_t)s * E->stride + k] = PIDX(dpart, pt->npair - 1); b->npair++;
		}
	}
	if(rows_seen >= 256){      /* the share of the rows that became pairs: follows the ranges down by a tenth per range at most, up at once */
		const double r = (double)b->npair / (double)rows_seen, keep = E->pair_row_ratio * 0.9;
		E->pair_row_ratio = r > keep ? r : keep;
	}
	E->t_cq[3] += now_s() - tp0;
}

/* the alignment items of a part: the best strand of every pair whose chain passes -r (wtzmo.c:913-914); a pure function of the summaries */
static void part_plan_items(eng_t *E, part_t *b){
	const wtz_params_c *P = &E->P;
	b->item_of = (uint32_t*)hx_realloc(b->item_of, 4 * ((size_t)b->npair + 1));
	b->it_pair = (uint32_t*)hx_realloc(b->it_pair, 4 * ((size_t)b->npair + 1));
	b->it_dir = (uint8_t*)hx_realloc(b->it_dir, (size_t)b->npair + 1);
	b->nitem = 0;
	for(uint32_t i = 0; i < b->npair; i++){
		b->item_of[i] = 0xFFFFFFFFu;
		if(!E->do_align || !b->sum[i].gate) continue;
		const uint32_t dir = (b->sum[i].ovl[0] < b->sum[i].ovl[1]);
		if(b->sum[i].ovl[dir] < P->ztot) continue;
		b->item_of[i] = b->nitem; b->it_pair[b->nitem] = i; b->it_dir[b->nitem] = (uint8_t)dir; b->nitem++;
	}
}
/* the page-locked CIGAR text buffer of the part for this range (two alternate: the records of the previous range may still be
 * waiting for the writer thread inside the other one); 0 = cannot allocate */
static int part_text_buffer(part_t *b, uint64_t tot){
	const int sel = (b->cig_sel ^= 1);
	b->cig_ext = b->ext_base >= 0 ? b->ext_base + sel : -1;
	{ const double tw0 = now_s(); out_wait_ext(b->cig_ext); b->t_io0 += now_s() - tw0; }
	if(tot > b->capcigs[sel]){
		uint64_t cap = b->capcigs[sel] ? b->capcigs[sel] : ((uint64_t)16 << 20); while(cap < tot) cap += cap / 2;
		wtz_host_free(b->cigs[sel]); b->cigs[sel] = (char*)wtz_host_alloc(cap + 1); b->capcigs[sel] = cap;
		if(!b->cigs[sel]){ fprintf(stderr, "[wtzmo-mi355x] cannot allocate %llu bytes of page-locked memory for the CIGAR text\n", (unsigned long long)cap); return 0; }
	}
	b->cig = b->cigs[sel];
	return 1;
}
/* the CIGAR text of the part's last range is on its way to the host (wtz_fetch_cigar_text_begin): wait for it */
static void part_text_wait(part_t *pt, double *t_acc){
	if(!pt->text_pending || pt->ctx == NULL) return;
	const double t0 = now_s();
	const int rc = wtz_fetch_cigar_text_end(pt->ctx);
	if(t_acc) *t_acc += now_s() - t0;
	pt->text_pending = 0;
	if(rc != WTZ_OK){ fprintf(stderr, " -- wtz_fetch_cigar_text_end failed: %s --\n", wtz_last_error()); DIE_NOW(); }
}
/* no lock held: the speculative device stages of ONE part's pairs on its context. 1 = scratch pool too small, nothing changed */
static int part_stages(eng_t *E, part_t *b){
	const wtz_params_c *P = &E->P;
	int rc;
	b->nitem = 0; b->ncig = 0;
	if(b->npair == 0) return 0;
	b->sum = (wtz_pair_summary_t*)hx_realloc(b->sum, sizeof(wtz_pair_summary_t) * (b->npair + 1));
	{ const double tc0 = now_s(); rc = wtz_pairs_seed(b->ctx, b->pq, b->pc, b->npair, b->sum); b->t_call[1] += now_s() - tc0; } TRY_WTZ(rc, "wtz_pairs_seed");
	if(!P->dot_matrix){
		b->box_off = (uint64_t*)hx_realloc(b->box_off, 8 * ((size_t)b->npair * 2 + 1));
		uint64_t nb = 0;
		for(uint32_t i = 0; i < b->npair; i++) for(int d = 0; d < 2; d++){ b->box_off[(size_t)i * 2 + d] = nb; nb += b->sum[i].nwin[d]; }
		b->box_off[(size_t)b->npair * 2] = nb; b->nbox = nb;
		if(nb > b->capbox){ b->capbox = nb; b->boxes = (wtz_winbox_t*)hx_realloc(b->boxes, sizeof(wtz_winbox_t) * nb); }
		{ const double tc0 = now_s(); rc = wtz_pairs_windows(b->ctx, b->boxes, nb); b->t_call[2] += now_s() - tc0; } TRY_WTZ(rc, "wtz_pairs_windows");
		part_plan_items(E, b);
		if(b->nitem){
			b->aln = (wtz_aln_result_t*)hx_realloc(b->aln, sizeof(wtz_aln_result_t) * b->nitem);
			{ const double tc0 = now_s(); rc = wtz_pairs_align(b->ctx, b->it_pair, b->it_dir, b->nitem, b->aln); b->t_call[3] += now_s() - tc0; } TRY_WTZ(rc, "wtz_pairs_align");
			uint64_t tot = 0; for(uint32_t i = 0; i < b->nitem; i++) tot += b->aln[i].text_len;
			if(E->binary_out) tot = 0;              /* binary records carry no CIGAR column (what `cut -f1-16` drops in the zmo pipeline): 3 GB per configs[2] step stay on the device */
			else if(g_dist.rank > 0 && g_dist.send_dev){      /* this rank only forwards the text: it never leaves the device on this side */
				const double tc0 = now_s(); b->cig_dev = NULL; rc = wtz_cigar_text_device(b->ctx, tot, &b->cig_dev); b->t_call[4] += now_s() - tc0; TRY_WTZ(rc, "wtz_cigar_text_device");
			} else {
				if(!part_text_buffer(b, tot)) return WTZ_ST_AGAIN;
				if(g_dist.world == 1){      /* the copy runs beside what the context does next: part_text_wait() before anybody reads the text */
					const double tc0 = now_s(); rc = wtz_fetch_cigar_text_begin(b->ctx, b->cig, tot); b->t_call[4] += now_s() - tc0; TRY_WTZ(rc, "wtz_fetch_cigar_text_begin"); b->text_pending = 1;
				} else
				{ const double tc0 = now_s(); rc = wtz_fetch_cigar_text(b->ctx, b->cig, tot); b->t_call[4] += now_s() - tc0; } TRY_WTZ(rc, "wtz_fetch_cigar_text");     /* rendered on the device */
			}
			b->ncig = tot;
		}
	}
	return 0;
}

static void *part_main(void *arg){ part_t *pt = (part_t*)arg; pt->again = part_stages(pt->E, pt); return NULL; }
/* all parts of the range side by side (one host thread per extra part); 1 = some part's scratch pool was too small */
static int gpu_stages_ranks(eng_t *E, batch_t *b);
static int gpu_stages(eng_t *E, batch_t *b){
	if(g_dist.world > 1) return gpu_stages_ranks(E, b);
	pthread_t th[16];
	for(uint32_t d = 1; d < b->nparts; d++){ b->parts[d].E = E; if(pthread_create(&th[d], NULL, part_main, &b->parts[d]) != 0){ fprintf(stderr, " -- cannot start a device thread --\n"); DIE_NOW(); } }
	b->parts[0].E = E; b->parts[0].again = part_stages(E, &b->parts[0]);
	int again = b->parts[0].again;
	for(uint32_t d = 1; d < b->nparts; d++){ pthread_join(th[d], NULL); again |= b->parts[d].again; }
	b->nitem = 0; for(uint32_t d = 0; d < b->nparts; d++) b->nitem += b->parts[d].nitem;
	return again;
}

/* rank 0, at a boundary of the protocol (every other rank is waiting for the next request): end all ranks */
static void ranks_abort(const char *why){
	fprintf(stderr, " -- %s: ending all %d ranks --\n", why, g_dist.world);
	wtz_dist_hdr_t h; memset(&h, 0, sizeof h); h.cmd = WTZ_CMD_ABORT;
	g_dist.bcast(&h, sizeof h);
	DIE_NOW();
}
/* test hook: WTZ_RANK_FAIL_AT="<rank>:<n>" makes that rank's n-th device-stage request fail (the in-band abort path must end every rank with exit code 1) */
static int rank_fail_injected(void){
	static int rank = -2, at = 0, seen = 0;
	if(rank == -2){ const char *e = getenv("WTZ_RANK_FAIL_AT"); rank = -1; if(e && sscanf(e, "%d:%d", &rank, &at) != 2) rank = -1; }
	if(rank != g_dist.rank) return 0;
	if(++seen != at) return 0;
	fprintf(stderr, " -- rank %d: injected failure of request %d (WTZ_RANK_FAIL_AT) --\n", g_dist.rank, at);
	return 1;
}

static uint8_t *part_xbuf(part_t *pt, uint64_t n){
	if(n > pt->xcap){ pt->xcap = n + n / 4 + 4096; pt->xbuf = (uint8_t*)hx_realloc(pt->xbuf, pt->xcap); }
	return pt->xbuf;
}
/* rank 0: part r of the range is computed by rank r (part 0 here, meanwhile) */
static int gpu_stages_ranks(eng_t *E, batch_t *b){
	const int dm = E->P.dot_matrix;
	wtz_dist_hdr_t h; memset(&h, 0, sizeof h); h.cmd = WTZ_CMD_PAIRS;
	for(uint32_t r = 0; r < b->nparts; r++) h.count[r] = b->parts[r].npair;
	g_dist.bcast(&h, sizeof h);
	for(uint32_t r = 1; r < b->nparts; r++){
		part_t *pt = &b->parts[r];
		if(pt->npair){      /* queries | candidates in one message */
			uint8_t *x = part_xbuf(pt, 8 * (uint64_t)pt->npair);
			memcpy(x, pt->pq, 4 * (size_t)pt->npair); memcpy(x + 4 * (size_t)pt->npair, pt->pc, 4 * (size_t)pt->npair);
			g_dist.send(x, 8 * (uint64_t)pt->npair, (int)r);
		}
	}
	b->parts[0].E = E;
	int st = b->parts[0].again = rank_fail_injected() ? WTZ_ST_FAILED : part_stages(E, &b->parts[0]);
	int failed = st == WTZ_ST_FAILED ? 0 : -1;      /* the rank that failed (its replies still complete the round) */
	for(uint32_t r = 1; r < b->nparts; r++){
		part_t *pt = &b->parts[r];
		uint64_t rh[4]; g_dist.recv(rh, sizeof rh, (int)r);
		pt->nitem = 0; pt->ncig = 0; pt->again = (int)rh[0];
		if(rh[0] == WTZ_ST_FAILED){ if(failed < 0) failed = (int)r; continue; }
		if(rh[0]){ if(st == WTZ_ST_OK) st = WTZ_ST_AGAIN; continue; }
		if(pt->npair == 0) continue;
		/* the reply proper: summaries | window boxes | alignment results in ONE message whose size the status word's counts give (round 5: three messages);
		 * the CIGAR text follows as a message of its own - it comes straight out of the peer's device memory */
		const uint64_t nb_r = dm ? 0 : rh[1], ni_r = dm ? 0 : rh[2];
		const uint64_t sz_sum = sizeof(wtz_pair_summary_t) * (uint64_t)pt->npair, sz_box = sizeof(wtz_winbox_t) * nb_r, sz_aln = sizeof(wtz_aln_result_t) * ni_r;
		uint8_t *x = part_xbuf(pt, sz_sum + sz_box + sz_aln);
		g_dist.recv(x, sz_sum + sz_box + sz_aln, (int)r);
		pt->sum = (wtz_pair_summary_t*)hx_realloc(pt->sum, sizeof(wtz_pair_summary_t) * (pt->npair + 1));
		memcpy(pt->sum, x, sz_sum);
		if(dm) continue;
		pt->box_off = (uint64_t*)hx_realloc(pt->box_off, 8 * ((size_t)pt->npair * 2 + 1));
		uint64_t nb = 0;
		for(uint32_t i = 0; i < pt->npair; i++) for(int d = 0; d < 2; d++){ pt->box_off[(size_t)i * 2 + d] = nb; nb += pt->sum[i].nwin[d]; }
		pt->box_off[(size_t)pt->npair * 2] = nb; pt->nbox = nb;
		if(nb != rh[1]){ fprintf(stderr, " -- rank %u reports %llu windows, its summaries say %llu --\n", r, (unsigned long long)rh[1], (unsigned long long)nb); DIE_NOW(); }
		if(nb > pt->capbox){ pt->capbox = nb; pt->boxes = (wtz_winbox_t*)hx_realloc(pt->boxes, sizeof(wtz_winbox_t) * nb); }
		if(nb) memcpy(pt->boxes, x + sz_sum, sz_box);
		part_plan_items(E, pt);
		if(pt->nitem != rh[2]){ fprintf(stderr, " -- rank %u aligned %llu items, the plan has %u --\n", r, (unsigned long long)rh[2], pt->nitem); DIE_NOW(); }
		if(pt->nitem){
			pt->aln = (wtz_aln_result_t*)hx_realloc(pt->aln, sizeof(wtz_aln_result_t) * pt->nitem);
			memcpy(pt->aln, x + sz_sum + sz_box, sz_aln);
			if(!part_text_buffer(pt, rh[3])) DIE_NOW();
			if(rh[3]) g_dist.recv(pt->cig, rh[3], (int)r);
			pt->ncig = rh[3];
		}
	}
	if(failed >= 0){ char why[64]; snprintf(why, sizeof why, "rank %d failed in the device stages of a range", failed); ranks_abort(why); }
	b->nitem = 0; for(uint32_t d = 0; d < b->nparts; d++) b->nitem += b->parts[d].nitem;
	return st;
}

/* the z-mer index of a batch on one context: `cl` (zbatch) = the candidate-side reads of this device for the batch, `ql` (several parts) = the batch's queries */
static int zbuild_part(wtz_ctx_t *ctx, int have_c, const uint32_t *cl, uint32_t ncl, int have_q, const uint32_t *ql, uint32_t nql){
	int rc = WTZ_OK;
	if(have_c) rc = wtz_zindex_build_subset(ctx, cl, ncl);
	if(rc == WTZ_OK && have_q) rc = wtz_zindex_build_queries(ctx, ql, nql);
	return rc;
}

/* ranks > 0: serve rank 0's requests with this GPU until the overlap phase is over.  A failure here never ends the process on the spot: it is reported in the
 * status word of the next reply (the requests in between are answered with empty payloads), and rank 0 ends all ranks with WTZ_CMD_ABORT. */
static void remote_loop(eng_t *E, part_t *pt){
	const int dm = E->P.dot_matrix; const int me = g_dist.rank;
	int failed = 0;
#define REMOTE_TRY(rc, what) do { if((rc) != WTZ_OK && !failed){ failed = 1; fprintf(stderr, " -- rank %d: %s failed: %s --\n", me, what, wtz_last_error()); } } while(0)
	pt->E = E;
	for(;;){
		wtz_dist_hdr_t h; g_dist.bcast(&h, sizeof h);
		if(h.cmd == WTZ_CMD_DONE) break;
		if(h.cmd == WTZ_CMD_ABORT){ fprintf(stderr, " -- rank %d: abort requested by rank 0 --\n", me); DIE_NOW(); }
		const uint32_t n = (uint32_t)h.count[me];
		if(h.cmd == WTZ_CMD_PAIRS){
			if(n > pt->cappair){ pt->cappair = n; pt->pq = (uint32_t*)hx_realloc(pt->pq, 4 * (size_t)n); pt->pc = (uint32_t*)hx_realloc(pt->pc, 4 * (size_t)n); }
			if(n){ uint8_t *x = part_xbuf(pt, 8 * (uint64_t)n); g_dist.recv(x, 8 * (uint64_t)n, 0); memcpy(pt->pq, x, 4 * (size_t)n); memcpy(pt->pc, x + 4 * (size_t)n, 4 * (size_t)n); }
			pt->npair = n;
			if(rank_fail_injected()) failed = 1;
			const int again = failed ? WTZ_ST_FAILED : part_stages(E, pt);
			if(again == WTZ_ST_FAILED) failed = 1;
			uint64_t rh[4] = { (uint64_t)again, pt->nbox, pt->nitem, pt->ncig };
			if(n == 0 || dm || again){ rh[1] = 0; rh[2] = 0; rh[3] = 0; }
			g_dist.send(rh, sizeof rh, 0);
			if(again || n == 0) continue;
			{
				const uint64_t sz_sum = sizeof(wtz_pair_summary_t) * (uint64_t)n, sz_box = dm ? 0 : sizeof(wtz_winbox_t) * pt->nbox, sz_aln = dm ? 0 : sizeof(wtz_aln_result_t) * (uint64_t)pt->nitem;
				uint8_t *x = part_xbuf(pt, sz_sum + sz_box + sz_aln);
				memcpy(x, pt->sum, sz_sum); if(sz_box) memcpy(x + sz_sum, pt->boxes, sz_box); if(sz_aln) memcpy(x + sz_sum + sz_box, pt->aln, sz_aln);
				g_dist.send(x, sz_sum + sz_box + sz_aln, 0);
			}
			if(!dm && pt->nitem && pt->ncig){ if(g_dist.send_dev) g_dist.send_dev(pt->cig_dev, pt->ncig, 0); else g_dist.send(pt->cig, pt->ncig, 0); }
		} else if(h.cmd == WTZ_CMD_ZIDX){
			const uint32_t nql = (uint32_t)h.arg[0]; const int have_c = h.arg[1] != 0;
			uint32_t *ql = (uint32_t*)hx_realloc(NULL, 4 * ((size_t)nql + 1)), *cl = (uint32_t*)hx_realloc(NULL, 4 * ((size_t)n + 1));
			if(nql) g_dist.bcast(ql, 4 * (uint64_t)nql);
			if(have_c && n) g_dist.recv(cl, 4 * (uint64_t)n, 0);
			if(!failed){ int rc = zbuild_part(pt->ctx, have_c, cl, n, 1, ql, nql); REMOTE_TRY(rc, "the z-mer index of the batch"); }
			free(ql); free(cl);
		} else if(h.cmd == WTZ_CMD_CAND_BEGIN){
			if(n > pt->cq_cap){ pt->cq_cap = n; pt->cq_ids = (uint32_t*)hx_realloc(pt->cq_ids, 4 * (size_t)n); pt->cq_nr = (uint32_t*)hx_realloc(pt->cq_nr, 4 * (size_t)n); pt->cq_rows = (uint64_t*)hx_realloc(pt->cq_rows, (size_t)n * E->stride * 8); }
			pt->cq_n = n;
			if(n){ g_dist.recv(pt->cq_ids, 4 * (uint64_t)n, 0); memset(pt->cq_rows, 0, (size_t)n * E->stride * 8); memset(pt->cq_nr, 0, 4 * (size_t)n); }
			if(!failed){ int rc = wtz_candidates_begin(pt->ctx, pt->cq_ids, n, pt->cq_rows, pt->cq_nr); REMOTE_TRY(rc, "wtz_candidates_begin"); }
		} else if(h.cmd == WTZ_CMD_CAND_END){
			if(!failed){ int rc = wtz_candidates_end(pt->ctx, pt->cq_rows, pt->cq_nr); REMOTE_TRY(rc, "wtz_candidates_end"); }
			{   /* status | rows | counts in one message of a size rank 0 knows (a failed rank sends the same size with the status set) */
				const uint64_t sz_rows = (uint64_t)pt->cq_n * E->stride * 8, sz_nr = 4 * (uint64_t)pt->cq_n;
				uint8_t *x = part_xbuf(pt, 8 + sz_rows + sz_nr);
				const uint64_t st = failed ? WTZ_ST_FAILED : WTZ_ST_OK; memcpy(x, &st, 8);
				if(pt->cq_n){ memcpy(x + 8, pt->cq_rows, sz_rows); memcpy(x + 8 + sz_rows, pt->cq_nr, sz_nr); }
				g_dist.send(x, 8 + sz_rows + sz_nr, 0);
			}
		} else if(h.cmd == WTZ_CMD_GRP_BEGIN){
			/* sharded index: every rank answers every query of the request with the groups of its shard */
			const uint32_t nq = (uint32_t)h.count[0];
			if(nq > pt->cq_cap){ pt->cq_cap = nq; pt->cq_ids = (uint32_t*)hx_realloc(pt->cq_ids, 4 * (size_t)nq); pt->cq_nr = (uint32_t*)hx_realloc(pt->cq_nr, 4 * (size_t)nq); pt->cq_rows = (uint64_t*)hx_realloc(pt->cq_rows, (size_t)nq * E->stride * 8); }
			pt->cq_n = nq;
			if(nq) g_dist.bcast(pt->cq_ids, 4 * (uint64_t)nq);
			if(!failed){ int rc = wtz_candidate_groups_begin(pt->ctx, pt->cq_ids, nq); REMOTE_TRY(rc, "wtz_candidate_groups_begin"); }
		} else if(h.cmd == WTZ_CMD_GRP_END){
			uint64_t tot = 0; uint64_t *gr = NULL;
			if(!failed){ int rc = wtz_candidate_groups_end(pt->ctx, pt->cq_nr); REMOTE_TRY(rc, "wtz_candidate_groups_end"); }
			if(!failed){
				for(uint32_t k = 0; k < pt->cq_n; k++) tot += pt->cq_nr[k];
				gr = (uint64_t*)hx_realloc(NULL, 8 * (tot + 1));
				int rc = wtz_candidate_groups_fetch(pt->ctx, gr, tot); REMOTE_TRY(rc, "wtz_candidate_groups_fetch");
			}
			{   /* status | group counts in one message of a known size, then the groups */
				uint8_t *x = part_xbuf(pt, 8 + 4 * (uint64_t)pt->cq_n);
				const uint64_t st = failed ? WTZ_ST_FAILED : WTZ_ST_OK; memcpy(x, &st, 8);
				if(pt->cq_n){ if(failed) memset(x + 8, 0, 4 * (size_t)pt->cq_n); else memcpy(x + 8, pt->cq_nr, 4 * (size_t)pt->cq_n); }
				g_dist.send(x, 8 + 4 * (uint64_t)pt->cq_n, 0);
				if(!failed && tot) g_dist.send(gr, 8 * tot, 0);
			}
			free(gr);
		} else { fprintf(stderr, " -- rank %d: unknown request %llu --\n", me, (unsigned long long)h.cmd); DIE_NOW(); }
	}
#undef REMOTE_TRY
}

/* Candidate search (A3) depends on the read and the index only, so the next batch's request can be in flight while this
 * batch is committed on the host.  The next queries are chosen with the masks as they are NOW; whoever the commit masks or
 * saturates meanwhile is dropped / demoted when the batch is formed - the batch composition is free (any batch size gives
 * the same output), only the query ORDER and the one-query masking lag are part of the contract. */
static void shard_candidates_begin(eng_t *E, const uint32_t *ids, uint32_t n);
static void ranks_abort(const char *why);
static void prefetch_begin(eng_t *E, batch_t *b){
	if(E->n_workers != 1 || E->rows_all || E->cursor >= E->qend) return;
	const uint32_t B = E->B;
	if(B > b->pf_cap){ b->pf_cap = B; b->pf_ids = (uint32_t*)hx_realloc(b->pf_ids, 4 * (size_t)B); b->pf_rows = (uint64_t*)hx_realloc(b->pf_rows, (size_t)B * E->stride * 8); b->pf_nr = (uint32_t*)hx_realloc(b->pf_nr, 4 * (size_t)B); }
	uint32_t j = E->cursor, n = 0;
	for(; j < E->qend && n < B; j++){
		if((j % E->n_job) != E->i_job) continue;
		if(E->masked[j]) continue;
		if(E->rdcovs[j] >= nbest_of(E, j)) continue;        /* already saturated: needs no candidates; it is re-examined when the batch is formed */
		b->pf_ids[n] = j; b->pf_nr[n] = 0; n++;
	}
	b->pf_n = n; b->pf_cursor_end = j;
	if(n == 0) return;
	memset(b->pf_rows, 0, (size_t)n * E->stride * 8);
	if(E->shard) shard_candidates_begin(E, b->pf_ids, n);
	else if(b->nparts == 1){
		int rc = wtz_candidates_begin(b->ctx, b->pf_ids, n, b->pf_rows, b->pf_nr); DIE_WTZ(rc, "wtz_candidates_begin");
	} else {
		/* the seed lookup is pure per query: query k of the request goes to device / rank k % nparts (the launches return at once) */
		int local_failed = 0;
		if(g_dist.world > 1){
			wtz_dist_hdr_t h; memset(&h, 0, sizeof h); h.cmd = WTZ_CMD_CAND_BEGIN;
			for(uint32_t d = 0; d < b->nparts; d++) h.count[d] = (n + b->nparts - 1 - d) / b->nparts;
			g_dist.bcast(&h, sizeof h);
		}
		for(uint32_t d = 0; d < b->nparts; d++){
			part_t *pt = &b->parts[d];
			const uint32_t m = (n + b->nparts - 1 - d) / b->nparts;
			if(m > pt->cq_cap){ pt->cq_cap = m; pt->cq_ids = (uint32_t*)hx_realloc(pt->cq_ids, 4 * (size_t)m); pt->cq_nr = (uint32_t*)hx_realloc(pt->cq_nr, 4 * (size_t)m); pt->cq_rows = (uint64_t*)hx_realloc(pt->cq_rows, (size_t)m * E->stride * 8); }
			pt->cq_n = m;
			for(uint32_t k = 0; k < m; k++){ pt->cq_ids[k] = b->pf_ids[(size_t)k * b->nparts + d]; pt->cq_nr[k] = 0; }
			if(m) memset(pt->cq_rows, 0, (size_t)m * E->stride * 8);
			if(pt->remote){ if(m) g_dist.send(pt->cq_ids, 4 * (uint64_t)m, pt->remote); continue; }
			int rc = wtz_candidates_begin(pt->ctx, pt->cq_ids, m, pt->cq_rows, pt->cq_nr);
			if(rc != WTZ_OK && g_dist.world > 1){ fprintf(stderr, " -- rank 0: wtz_candidates_begin failed: %s --\n", wtz_last_error()); local_failed = 1; continue; }
			DIE_WTZ(rc, "wtz_candidates_begin");
		}
		if(local_failed) ranks_abort("rank 0 failed in the seed lookup");      /* every rank has its ids: the request is complete, the next thing they read is a header */
	}
	b->pf_inflight = 1;
}


/* ---- k-mer index sharded by read-id range over the devices (--shard-index; include/wtzmo_hip.h).  One exchange: the shards' distinct k-mers
 * with their counts are merged on the host, the frequency filter and the automatic cutoff (wtzmo.c:380-405) use the TOTAL counts, so the
 * tables together hold exactly the k-mers of the unsharded index and every seed run is the unsharded run cut at the shard borders. ---- */
#define SHARD_N(E) (g_dist.world > 1 ? (uint32_t)g_dist.world : (E)->ndev)      /* one shard per device of this process, or per rank (one process per GPU) */
static void shard_range(const eng_t *E, uint32_t n_rd, uint32_t d, uint32_t *b, uint32_t *e){
	const uint32_t N = SHARD_N(E), per = (n_rd + N - 1) / N;      /* contiguous id ranges like the reference's -G parts (wtzmo.c:1281-1285) */
	*b = d * per < n_rd ? d * per : n_rd; *e = (d + 1) * per < n_rd ? (d + 1) * per : n_rd;
}
static void shard_index_build(eng_t *E, uint32_t n_rd, uint32_t *K_io){
	const uint32_t N = SHARD_N(E); const int ranks = g_dist.world > 1, me = g_dist.rank;
	uint64_t nd[WTZ_DIST_MAX], nocc[WTZ_DIST_MAX], *km[WTZ_DIST_MAX]; uint32_t *lc[WTZ_DIST_MAX], *tc[WTZ_DIST_MAX];
	memset(km, 0, sizeof km); memset(lc, 0, sizeof lc); memset(tc, 0, sizeof tc);
	for(uint32_t d = 0; d < N; d++){
		if(ranks && (int)d != me) continue;                         /* the other ranks count their own shards */
		wtz_ctx_t *cx = ranks ? E->ctx : E->ctxs[d];
		uint32_t b, e; shard_range(E, n_rd, d, &b, &e);
		int rc = wtz_index_count(cx, b, e, &nd[d], &nocc[d]); DIE_WTZ(rc, "wtz_index_count");
		km[d] = (uint64_t*)hx_realloc(NULL, 8 * (nd[d] + 1)); lc[d] = (uint32_t*)hx_realloc(NULL, 4 * (nd[d] + 1)); tc[d] = (uint32_t*)hx_realloc(NULL, 4 * (nd[d] + 1));
		rc = wtz_index_counts_fetch(cx, km[d], lc[d]); DIE_WTZ(rc, "wtz_index_counts_fetch");
	}
	uint64_t kbuf[4] = {0, 0, 0, 0};      /* K, occurrences, distinct, table entries: rank 0 -> all */
	if(ranks && me != 0){
		/* the one exchange of the build: this shard's (k-mer, count) list to rank 0, the total counts of the same k-mers back */
		uint64_t hd[2] = { nd[me], nocc[me] };
		g_dist.send(hd, sizeof hd, 0);
		if(nd[me]){ g_dist.send(km[me], 8 * nd[me], 0); g_dist.send(lc[me], 4 * nd[me], 0); g_dist.recv(tc[me], 4 * nd[me], 0); }
		g_dist.bcast(kbuf, sizeof kbuf);
		uint64_t kept = 0; int rc = wtz_index_finish(E->ctx, tc[me], (uint32_t)kbuf[0], &kept); DIE_WTZ(rc, "wtz_index_finish");
		*K_io = (uint32_t)kbuf[0];
		free(km[me]); free(lc[me]); free(tc[me]);
		return;
	}
	if(ranks) for(uint32_t r = 1; r < N; r++){
		uint64_t hd[2]; g_dist.recv(hd, sizeof hd, (int)r); nd[r] = hd[0]; nocc[r] = hd[1];
		km[r] = (uint64_t*)hx_realloc(NULL, 8 * (nd[r] + 1)); lc[r] = (uint32_t*)hx_realloc(NULL, 4 * (nd[r] + 1)); tc[r] = (uint32_t*)hx_realloc(NULL, 4 * (nd[r] + 1));
		if(nd[r]){ g_dist.recv(km[r], 8 * nd[r], (int)r); g_dist.recv(lc[r], 4 * nd[r], (int)r); }
	}
	/* N-way merge of the ascending lists: total count per k-mer, ktyp = distinct k-mers, ktot = sum of the saturating counts (wtzmo.c:276, 380-388) */
	uint64_t pos[WTZ_DIST_MAX], ktyp = 0, ktot = 0; memset(pos, 0, sizeof pos);
	for(;;){
		uint64_t m = ~0ull; int any = 0;
		for(uint32_t d = 0; d < N; d++) if(pos[d] < nd[d]){ if(!any || km[d][pos[d]] < m) m = km[d][pos[d]]; any = 1; }
		if(!any) break;
		uint64_t tot = 0;
		for(uint32_t d = 0; d < N; d++) if(pos[d] < nd[d] && km[d][pos[d]] == m) tot += lc[d][pos[d]];
		const uint32_t t32 = tot > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)tot;
		for(uint32_t d = 0; d < N; d++) if(pos[d] < nd[d] && km[d][pos[d]] == m){ tc[d][pos[d]] = t32; pos[d]++; }
		ktyp++; ktot += tot > 0xFFFFu ? 0xFFFFu : tot;
	}
	uint32_t K = *K_io;
	if(K < 2){ uint32_t kavg = (uint32_t)(ktot / (ktyp + 1)); if(kavg < 20) kavg = 20; K = kavg * 5; }       /* wtzmo.c:380-393 */
	*K_io = K;
	uint64_t occ = 0, kept_all = 0;
	for(uint32_t d = 0; d < N; d++){
		occ += nocc[d];
		if(ranks && d){ if(nd[d]) g_dist.send(tc[d], 4 * nd[d], (int)d); }
		else { uint64_t kept = 0; int rc = wtz_index_finish(ranks ? E->ctx : E->ctxs[d], tc[d], K, &kept); DIE_WTZ(rc, "wtz_index_finish"); kept_all += kept; }
		free(km[d]); free(lc[d]); free(tc[d]);
	}
	if(ranks){ kbuf[0] = K; kbuf[1] = occ; kbuf[2] = ktyp; g_dist.bcast(kbuf, sizeof kbuf); }
	fprintf(stderr, "[wtzmo-mi355x] index in %u shards by read id%s: %llu k-mer occurrences, %llu distinct, %llu table entries%s, cutoff %u\n", N, ranks ? " (one per rank)" : "",
		(unsigned long long)occ, (unsigned long long)ktyp, (unsigned long long)kept_all, ranks ? " in shard 0" : " over the shards", K);
}
/* candidate heaps of n queries against the sharded index: every shard answers every query with its (read, strand) groups (key order); the
 * shards are ascending id ranges, so their lists in shard order are the unsharded group list, and the heap replay runs over that */
static void shard_candidates_begin(eng_t *E, const uint32_t *ids, uint32_t n){
	if(g_dist.world > 1){
		wtz_dist_hdr_t h; memset(&h, 0, sizeof h); h.cmd = WTZ_CMD_GRP_BEGIN; h.count[0] = n;
		g_dist.bcast(&h, sizeof h);
		if(n) g_dist.bcast((void*)ids, 4 * (uint64_t)n);
		int rc = wtz_candidate_groups_begin(E->ctx, ids, n); DIE_WTZ(rc, "wtz_candidate_groups_begin");
		return;
	}
	for(uint32_t d = 0; d < E->ndev; d++){ int rc = wtz_candidate_groups_begin(E->ctxs[d], ids, n); DIE_WTZ(rc, "wtz_candidate_groups_begin"); }
}
static void shard_candidates_end(eng_t *E, uint32_t n, uint64_t *rows, uint32_t *nr){
	const uint32_t N = SHARD_N(E); const int ranks = g_dist.world > 1;
	uint32_t *ng[WTZ_DIST_MAX]; uint64_t *gr[WTZ_DIST_MAX], tot[WTZ_DIST_MAX]; int failed = -1;
	if(ranks){ wtz_dist_hdr_t h; memset(&h, 0, sizeof h); h.cmd = WTZ_CMD_GRP_END; g_dist.bcast(&h, sizeof h); }
	for(uint32_t d = 0; d < N; d++){
		ng[d] = (uint32_t*)hx_realloc(NULL, 4 * ((size_t)n + 1));
		if(ranks && d){
			uint64_t st = 0;
			{ uint8_t *x = (uint8_t*)hx_realloc(NULL, 8 + 4 * (size_t)n); g_dist.recv(x, 8 + 4 * (uint64_t)n, (int)d); memcpy(&st, x, 8); if(n) memcpy(ng[d], x + 8, 4 * (size_t)n); free(x); }
			tot[d] = 0; gr[d] = NULL;
			if(st != WTZ_ST_OK){ if(failed < 0) failed = (int)d; memset(ng[d], 0, 4 * ((size_t)n + 1)); continue; }
			for(uint32_t k = 0; k < n; k++) tot[d] += ng[d][k];
			gr[d] = (uint64_t*)hx_realloc(NULL, 8 * (tot[d] + 1));
			if(tot[d]) g_dist.recv(gr[d], 8 * tot[d], (int)d);
			continue;
		}
		wtz_ctx_t *cx = ranks ? E->ctx : E->ctxs[d];
		int rc = wtz_candidate_groups_end(cx, ng[d]);
		tot[d] = 0; gr[d] = NULL;
		if(rc == WTZ_OK){
			for(uint32_t k = 0; k < n; k++) tot[d] += ng[d][k];
			gr[d] = (uint64_t*)hx_realloc(NULL, 8 * (tot[d] + 1));
			rc = wtz_candidate_groups_fetch(cx, gr[d], tot[d]);
		}
		if(rc != WTZ_OK && ranks){ fprintf(stderr, " -- rank 0: the candidate groups of its shard failed: %s --\n", wtz_last_error()); if(failed < 0) failed = 0; memset(ng[d], 0, 4 * ((size_t)n + 1)); tot[d] = 0; continue; }
		DIE_WTZ(rc, "wtz_candidate_groups_end / _fetch");
	}
	if(failed >= 0){ char why[64]; snprintf(why, sizeof why, "rank %d failed in the sharded seed lookup", failed); ranks_abort(why); }
	uint64_t off[WTZ_DIST_MAX], *join = NULL; size_t capj = 0; memset(off, 0, sizeof off);
	for(uint32_t k = 0; k < n; k++){
		size_t m = 0; for(uint32_t d = 0; d < N; d++) m += ng[d][k];
		if(m + 1 > capj){ capj = (m + 1) * 2; join = (uint64_t*)hx_realloc(join, 8 * capj); }
		m = 0; for(uint32_t d = 0; d < N; d++){ memcpy(join + m, gr[d] + off[d], 8 * (size_t)ng[d][k]); m += ng[d][k]; off[d] += ng[d][k]; }
		uint32_t hn = nr[k];
		wtz_cand_tail_host(join, (uint32_t)m, E->P.kovl, E->P.ncand, rows + (size_t)k * E->stride, &hn);
		nr[k] = hn;
	}
	free(join);
	for(uint32_t d = 0; d < N; d++){ free(ng[d]); free(gr[d]); }
}

/* slots [s0,s1): plan pairs, run the device stages, commit in query order when it is this batch's turn. A range whose
 * scratch demand exceeds the pool is split in two (the second half is planned after the first half is committed). */
/* a range came back with "scratch pool too small": was it the library's transient trace pool alone (its own budget, raised by now) on every local part that failed?
 * Then the bytes-per-pair estimate of the MAIN pool is not what was wrong and stays as it is (one under-estimated trace budget used to halve every later
 * range).  Parts on other ranks cannot be asked: they count as a main-pool failure. */
static int range_failed_on_traces_only(const batch_t *b){
	if(g_dist.world > 1) return 0;
	int seen = 0;
	for(uint32_t d = 0; d < b->nparts; d++){
		const part_t *pt = &b->parts[d];
		if(!pt->ctx || pt->npair == 0) continue;
		const int k = wtz_pool_failure_kind(pt->ctx);
		if(k == 1) return 0;
		if(k == 2) seen = 1;
	}
	return seen;
}
static void process_range(eng_t *E, batch_t *b, uint32_t s0, uint32_t s1){
	pthread_mutex_lock(&E->mu);
	plan_pairs(E, b, s0, s1);
	pthread_mutex_unlock(&E->mu);
	const double tg0 = now_s();
	const int again = gpu_stages(E, b);
	for(uint32_t d = 0; d < b->nparts; d++) part_text_wait(&b->parts[d], &b->parts[d].t_call[4]);      /* this path commits at once */
	const double tg1 = now_s();
	if(!again && b->npair){
		for(uint32_t d = 0; d < b->nparts; d++){
			wtz_pool_info_t pi; const part_t *pt = &b->parts[d];
			if(pt->npair == 0 || wtz_pool_info(pt->ctx, &pi) != WTZ_OK) continue;
			pthread_mutex_lock(&E->mu);
			E->main_cap = pi.main_cap * b->nparts;                  /* the range is dealt over nparts pools */
			const double bpp = (double)pi.main_used / (double)pt->npair;
			if(pt->npair >= 8 && bpp > E->bytes_per_pair) E->bytes_per_pair = bpp;
			else if(pt->npair >= 256 && E->bpp_decay > 0 && bpp < E->bytes_per_pair){      /* the longest reads come first: let the estimate follow the measurements down, 15 % per range at most */
				const double floor_ = E->bytes_per_pair * E->bpp_decay; E->bytes_per_pair = bpp > floor_ ? bpp : floor_; }
			pthread_mutex_unlock(&E->mu);
		}
	}
	if(again){
		pthread_mutex_lock(&E->mu); E->n_split++; if(!range_failed_on_traces_only(b)) E->bytes_per_pair = (E->bytes_per_pair > 0 ? E->bytes_per_pair : 1048576.0) * 2.0; pthread_mutex_unlock(&E->mu);
		if(s1 - s0 <= 1){ fprintf(stderr, " -- device scratch pool too small even for one query: %s (use --pool-gb) --\n", wtz_last_error()); DIE_NOW(); }
		fprintf(stderr, "[wtzmo-mi355x] scratch pool exhausted with %u queries in flight; splitting the batch\n", s1 - s0);
		const uint32_t mid = s0 + (s1 - s0) / 2;
		process_range(E, b, s0, mid);
		process_range(E, b, mid, s1);
		return;
	}
	if(s1 == b->nbq && !b->pf_inflight){ pthread_mutex_lock(&E->mu); prefetch_begin(E, b); pthread_mutex_unlock(&E->mu); }
	pthread_mutex_lock(&E->mu);
	E->t_gpu += tg1 - tg0;
	for(uint32_t d = 0; d < b->nparts; d++){ part_t *pt = &b->parts[d]; for(int k = 0; k < 5; k++){ if(d == 0) E->t_call[k] += pt->t_call[k]; pt->t_call[k] = 0; } E->t_io[0] += pt->t_io0; pt->t_io0 = 0; }
	while(!b->holds_turn && E->commit_seq != b->seq) pthread_cond_wait(&E->cv, &E->mu);
	b->holds_turn = 1;
	E->spec_pairs += b->npair; E->spec_items += b->nitem;
	/* commit in query order with the reference's one-query masking lag (wtzmo.c:1315-1333) */
	{ const double tc = now_s(); cq_spec_start(E, b, s0, s1, b->npair); E->t_spec_ctl += now_s() - tc; }
	for(uint32_t s = s0; s < s1; s++){
		if(s + 1 < s1) commit_prefetch(E, b, s + 1);
		if(!E->masked[b->bq[s]]){ const double tf = now_s(); flush_pending(E); E->t_flush += now_s() - tf; commit_query(E, b, s); }
		cq_spec_done(E, s);
	}
	{ const double tc = now_s(); cq_spec_stop(E); E->t_spec_ctl += now_s() - tc; }
	out_submit(&g_ow);      /* nothing of this batch stays in the chunk under construction (see out_wait_ext) */
	E->t_commit += now_s() - tg1;
	pthread_mutex_unlock(&E->mu);
}

/* end of the range that starts at slot s0: as many slots as fit the main scratch pool at the measured bytes per pair */
static uint32_t range_end(eng_t *E, batch_t *b, uint32_t s0){
	pthread_mutex_lock(&E->mu);
	/* before the first measurement: a small probe range (about 2000 pairs, never more than 1 MB per pair allows) - repeat-rich reads
	 * need tens of MB per pair where iid reads need a few hundred KB, and the longest reads come first */
	const double cap = E->main_cap ? (double)E->main_cap : 0.0;
	const double prior = cap / 2048.0 > 1048576.0 ? cap / 2048.0 : 1048576.0;
	const double bpp = E->bytes_per_pair > 0 ? E->bytes_per_pair : prior;
	const double row_ratio = E->pair_row_ratio;
	pthread_mutex_unlock(&E->mu);
	static double fill = -1.0;      /* WTZ_RANGE_FILL: the share of the main pool a range is planned to (candidate rows are an upper bound of its pairs) */
	/* measured at configs[2] (round 5, gpurun_out/r05r, r05s; every stage of a range ends in the tail of its slowest tasks, so fewer and larger ranges win until the
	 * pool overflows and a range is redone in halves): 0.7 -> 28 ranges 2.10 s, 1.0 -> 21 ranges 1.98 s, 1.4 -> 16 ranges 1.97 s (pool peak 96 of 128 GB),
	 * 1.8 -> one overflow, 2.11 s; with batches of 8 192 queries 1.4 -> 14 ranges 1.94 s */
	if(fill < 0){ const char *e = getenv("WTZ_RANGE_FILL"); fill = e ? atof(e) : 1.2; if(fill <= 0) fill = 1.2; }
	/* the budget is in PAIRS, the slots are cut by candidate ROWS: closed pairs (the other read of the pair came first), masked and saturated queries thin the rows out -
	 * to 0.9 at configs[2], to less than a tenth at the configs[3] shape (70x: every pair is found from both sides), where ranges held 17 000 pairs for a pool that holds
	 * 40 000.  The share measured on the ranges before (+ 25 %) scales the budget; an estimate that was too low ends in the split of the range like any other */
	double ratio = row_ratio * 1.25; if(ratio > 1.0 || ratio <= 0) ratio = 1.0;
	uint64_t budget = cap > 0 ? (uint64_t)(fill * cap / bpp / ratio) : ~0ull;
	if(budget < 16) budget = 16;
	uint32_t s1 = s0; uint64_t acc = 0;
	while(s1 < b->nbq && (s1 == s0 || acc + (b->want[s1] ? b->nrow[s1] : 0) <= budget)){ acc += b->want[s1] ? b->nrow[s1] : 0; s1++; }
	return s1;
}
/* the slots of a batch in ranges whose pairs (candidate rows are their upper bound) fit the main scratch pool at the measured bytes per pair */
static void process_batch_serial(eng_t *E, batch_t *b, uint32_t s0){
	while(s0 < b->nbq){
		const uint32_t s1 = range_end(E, b, s0);
		process_range(E, b, s0, s1);
		pthread_mutex_lock(&E->mu); E->n_ranges++; pthread_mutex_unlock(&E->mu);
		s0 = s1;
	}
}
/* Ranges PIPELINED: while the host commits range r (sequential by definition: 0.36 s of a 3.7 s configs[2] step) the device stages of range
 * r + 1 run on a helper thread.  Range r + 1 is therefore planned BEFORE range r is committed, i.e. against masks / closed pairs / coverage
 * one range older: every piece of order-dependent state only removes work, so the plan is a superset of what the commit will
 * ask for and the output cannot change - a few more speculative pairs are computed.  The finished range's result arrays are swapped into
 * the batch's spare set (cparts) so that the next range can fill `parts`.  A range that overflows the scratch pool is redone by the serial
 * path (halving), then the pipeline starts again behind it. */
typedef struct { eng_t *E; batch_t *b; int again; double t0, t1; pthread_t th; int running; char err[256]; } gpujob_t;
static void *gpujob_main(void *arg){ gpujob_t *j = (gpujob_t*)arg; j->t0 = now_s(); j->again = gpu_stages(j->E, j->b); j->t1 = now_s();
	if(j->again) snprintf(j->err, sizeof j->err, "%s", wtz_last_error());      /* wtz_last_error is thread-local: the text is this helper thread's */
	return NULL; }
static void gpujob_start(gpujob_t *j, eng_t *E, batch_t *b){ j->E = E; j->b = b; j->again = 0; if(pthread_create(&j->th, NULL, gpujob_main, j) != 0){ fprintf(stderr, " -- cannot start the device-stage thread --\n"); DIE_NOW(); } j->running = 1; }
static int gpujob_wait(gpujob_t *j){ if(j->running){ pthread_join(j->th, NULL); j->running = 0; } return j->again; }
#define SWAP_FIELD(T, a, b) do { T t_ = (a); (a) = (b); (b) = t_; } while(0)
static int batch_form(batch_t *b);
static void process_batch(eng_t *E, batch_t *b){
	static int overlap = -1;
	if(overlap < 0){ const char *e = getenv("WTZ_RANGE_OVERLAP"); overlap = e ? atoi(e) : 1; }
	if(!overlap || E->n_workers != 1 || b->nbq == 0){ b->cparts = b->parts; process_batch_serial(E, b, 0); return; }
	if(b->start_job == NULL) b->start_job = calloc(1, sizeof(gpujob_t));
	gpujob_t *job = (gpujob_t*)b->start_job;      /* lives in the batch: the batch before may have started our first range already */
	uint32_t s0 = 0, s1;
	if(b->started){ s1 = b->start_s1; b->started = 0; }      /* the batch before this one started our first range in front of its last commit */
	else {
		s1 = range_end(E, b, s0);
		pthread_mutex_lock(&E->mu); plan_pairs(E, b, s0, s1); pthread_mutex_unlock(&E->mu);
		gpujob_start(job, E, b);
	}
	for(;;){
		const int again = gpujob_wait(job);
		if(again){
			/* scratch pool exhausted: nothing of [s0, s1) is committed and nothing else is in flight: the serial path splits it */
			for(uint32_t d = 0; d < b->nparts; d++) part_text_wait(&b->parts[d], NULL);
			b->cparts = b->parts;
			pthread_mutex_lock(&E->mu); E->n_split++; if(!range_failed_on_traces_only(b)) E->bytes_per_pair = (E->bytes_per_pair > 0 ? E->bytes_per_pair : 1048576.0) * 2.0; pthread_mutex_unlock(&E->mu);
			if(s1 - s0 <= 1){ fprintf(stderr, " -- device scratch pool too small even for one query: %s (use --pool-gb) --\n", job->err); DIE_NOW(); }
			fprintf(stderr, "[wtzmo-mi355x] scratch pool exhausted with %u queries in flight; splitting the batch\n", s1 - s0);
			const uint32_t mid = s0 + (s1 - s0) / 2;
			process_range(E, b, s0, mid); process_range(E, b, mid, s1);
			pthread_mutex_lock(&E->mu); E->n_ranges++; pthread_mutex_unlock(&E->mu);
			s0 = s1;
			if(s0 >= b->nbq) break;
			s1 = range_end(E, b, s0);
			pthread_mutex_lock(&E->mu); plan_pairs(E, b, s0, s1); pthread_mutex_unlock(&E->mu);
			gpujob_start(job, E, b);
			continue;
		}
		const double tg0 = job->t0, tg1 = job->t1;
		const uint32_t r_npair = b->npair, r_nitem = b->nitem;
		if(b->npair){
			for(uint32_t d = 0; d < b->nparts; d++){
				wtz_pool_info_t pi; const part_t *pt = &b->parts[d];
				if(pt->npair == 0 || pt->ctx == NULL || wtz_pool_info(pt->ctx, &pi) != WTZ_OK) continue;
				pthread_mutex_lock(&E->mu);
				E->main_cap = pi.main_cap * b->nparts;
				const double bpp = (double)pi.main_used / (double)pt->npair;
				if(pt->npair >= 8 && bpp > E->bytes_per_pair) E->bytes_per_pair = bpp;
				else if(pt->npair >= 256 && E->bpp_decay > 0 && bpp < E->bytes_per_pair){ const double floor_ = E->bytes_per_pair * E->bpp_decay; E->bytes_per_pair = bpp > floor_ ? bpp : floor_; }
				pthread_mutex_unlock(&E->mu);
			}
		}
		/* the finished range's results move to the commit view */
		for(uint32_t d = 0; d < b->nparts; d++){
			part_t *p = &b->parts[d], *c = &b->spare[d];
			SWAP_FIELD(wtz_pair_summary_t*, p->sum, c->sum); SWAP_FIELD(uint64_t*, p->box_off, c->box_off); SWAP_FIELD(wtz_winbox_t*, p->boxes, c->boxes);
			SWAP_FIELD(uint64_t, p->capbox, c->capbox); SWAP_FIELD(uint64_t, p->nbox, c->nbox);
			SWAP_FIELD(uint32_t*, p->item_of, c->item_of); SWAP_FIELD(uint32_t*, p->it_pair, c->it_pair); SWAP_FIELD(uint8_t*, p->it_dir, c->it_dir); SWAP_FIELD(wtz_aln_result_t*, p->aln, c->aln);
			c->cig = p->cig; c->cig_ext = p->cig_ext; c->ncig = p->ncig; c->nitem = p->nitem; c->npair = p->npair;
			c->ctx = p->ctx; c->text_pending = p->text_pending; p->text_pending = 0;
		}
		b->cparts = b->spare;
		double t_gpu_call[5] = {0, 0, 0, 0, 0}, t_io0 = 0;
		for(uint32_t d = 0; d < b->nparts; d++){ part_t *pt = &b->parts[d]; for(int k = 0; k < 5; k++){ if(d == 0) t_gpu_call[k] += pt->t_call[k]; pt->t_call[k] = 0; } t_io0 += pt->t_io0; pt->t_io0 = 0; }
		/* next range: planned now (one range of staleness), computed while this one is committed */
		const uint32_t n0 = s1;
		uint32_t n1 = n0;
		if(n0 < b->nbq){
			n1 = range_end(E, b, n0);
			pthread_mutex_lock(&E->mu); plan_pairs(E, b, n0, n1); pthread_mutex_unlock(&E->mu);
			gpujob_start(job, E, b);
		} else {
			if(!b->pf_inflight){ pthread_mutex_lock(&E->mu); prefetch_begin(E, b); pthread_mutex_unlock(&E->mu); }
			batch_t *nb = b->alt;
			/* ... where that pays.  The commit about to run masks reads (the longest reads come first and contain many of the reads behind them), and whatever it masks
			 * inside the range started early was computed for nothing: E. coli shape with every boundary overlapped 6 790 queries planned for 3 017 used, 0.274 s per step
			 * against 0.242; configs[2] 1.81 s against 1.85.  So the next batch is FORMED here (its candidates are known: that is what the estimate needs; a slot masked
			 * later is skipped when its range is planned) and its first range is STARTED here only if
			 *     expected waste = share of the reads to come that the commit will mask (new masks per used query of this batch so far x the queries of its last range /
			 *                      the unmasked reads to come, at most 1) x the pairs of that first range x the step's device time per planned pair
			 *  <= what is hidden  = the queries of the last range x the step's commit time per used query (+ 3 ms of forming and planning).
			 * WTZ_BATCH_OVERLAP_GAIN scales the right side (default 1; 0 = never start early, 1e9 = always) */
			static double gain = -1.0; if(gain < 0){ const char *e = getenv("WTZ_BATCH_OVERLAP_GAIN"); gain = e ? atof(e) : 1.0; }
			if(nb && b->pf_inflight && !nb->formed && gain > 0){
				const double rate = (s0 >= 64 && b->used_queries >= 32) ? (double)(E->n_masked - b->masked_at_form) / (double)b->used_queries : E->last_mask_rate;
				uint64_t left = 0; for(uint32_t j = E->cursor; j < E->qend; j++) left += !E->masked[j];
				uint32_t n_last = 0; for(uint32_t s = s0; s < s1; s++) n_last += (b->want[s] && !E->masked[b->bq[s]]);
				double waste = rate < 0 ? 1.0 : rate * (double)n_last / (double)(left ? left : 1); if(waste > 1.0) waste = 1.0;
#define SWAP_PF(T, f) do { T t_ = b->f; b->f = nb->f; nb->f = t_; } while(0)
				SWAP_PF(int, pf_inflight); SWAP_PF(uint32_t*, pf_ids); SWAP_PF(uint32_t, pf_n); SWAP_PF(uint32_t, pf_cap); SWAP_PF(uint64_t*, pf_rows); SWAP_PF(uint32_t*, pf_nr); SWAP_PF(uint32_t, pf_cursor_end);
				for(uint32_t d = 0; d < b->nparts; d++){      /* several parts: the request's per-part arrays travel with it */
					part_t *pa = &b->parts[d], *pn = &nb->parts[d];
#define SWAP_CQ(T, f) do { T t_ = pa->f; pa->f = pn->f; pn->f = t_; } while(0)
					SWAP_CQ(uint32_t*, cq_ids); SWAP_CQ(uint32_t*, cq_nr); SWAP_CQ(uint64_t*, cq_rows); SWAP_CQ(uint32_t, cq_cap); SWAP_CQ(uint32_t, cq_n);
#undef SWAP_CQ
				}
#undef SWAP_PF
				if(batch_form(nb)){
					nb->formed = 1;
					if(nb->nbq){
						const uint32_t s1n = range_end(E, nb, 0);
						uint64_t rows = 0; for(uint32_t s = 0; s < s1n; s++) rows += nb->want[s] ? nb->nrow[s] : 0;
						pthread_mutex_lock(&E->mu);
						const double per_pair = E->spec_pairs ? E->t_gpu / (double)E->spec_pairs : -1.0, per_commit = E->used_queries ? E->t_commit / (double)E->used_queries : 0.0;
						pthread_mutex_unlock(&E->mu);
						const double cost = per_pair < 0 ? 1e30 : waste * (double)rows * per_pair, hidden = (double)n_last * per_commit + 0.003;
						const int early = gain >= 1e8 || cost <= gain * hidden;
						if(getenv("WTZ_BATCH_OVERLAP_TRACE")) fprintf(stderr, "[batch-overlap] batch %llu: %.2f new masks per used query, last range %u queries, %llu unmasked reads to come: share masked %.3f; first range of the next batch %llu pairs: expected waste %.1f ms against %.1f ms hidden -> %s\n",
							(unsigned long long)b->seq, rate, n_last, (unsigned long long)left, waste, (unsigned long long)rows, cost < 1e29 ? cost * 1e3 : -1.0, hidden * 1e3, early ? "started early" : "formed only");
						if(early){
							if(nb->start_job == NULL) nb->start_job = calloc(1, sizeof(gpujob_t));
							nb->start_s1 = s1n;
							pthread_mutex_lock(&E->mu); plan_pairs(E, nb, 0, nb->start_s1); pthread_mutex_unlock(&E->mu);
							gpujob_start((gpujob_t*)nb->start_job, E, nb);
							nb->started = 1;
						}
					}
				}
			}
		}
		for(uint32_t d = 0; d < b->nparts; d++) part_text_wait(&b->spare[d], &t_gpu_call[4]);      /* the finished range's text: its copy ran beside the planning above and the first kernels of the next range */
		const double tc0 = now_s();
		pthread_mutex_lock(&E->mu);
		E->t_gpu += tg1 - tg0;
		for(int k = 0; k < 5; k++) E->t_call[k] += t_gpu_call[k];
		E->t_io[0] += t_io0;
		while(!b->holds_turn && E->commit_seq != b->seq) pthread_cond_wait(&E->cv, &E->mu);
		b->holds_turn = 1;
		E->spec_pairs += r_npair; E->spec_items += r_nitem;
		{ const double tc = now_s(); cq_spec_start(E, b, s0, s1, r_npair); E->t_spec_ctl += now_s() - tc; }
		for(uint32_t s = s0; s < s1; s++){
			if(s + 1 < s1) commit_prefetch(E, b, s + 1);
			if(!E->masked[b->bq[s]]){ const double tf = now_s(); flush_pending(E); E->t_flush += now_s() - tf; commit_query(E, b, s); }
			cq_spec_done(E, s);
		}
		{ const double tc = now_s(); cq_spec_stop(E); E->t_spec_ctl += now_s() - tc; }
		out_submit(&g_ow);
		E->t_commit += now_s() - tc0;
		E->n_ranges++;
		pthread_mutex_unlock(&E->mu);
		if(n0 >= b->nbq) break;
		s0 = n0; s1 = n1;
	}
	b->cparts = b->parts;
}

/* both index builds of one more device (replicated indexes, --gpus) */
/* the device contexts - above all the hipMalloc of their scratch pools: ~35 ms per GB, 4.4 s for the default 128 GB - are created on a helper
 * thread while the main thread reads the FASTA (measured on wtgbo, E. coli shape: 5.4 s wall with the 128 GB pool created up front, 1.0 s with 16 GB) */
typedef struct { const wtz_params_c *P; uint64_t pool_bytes; int devs[8]; uint32_t ndev; wtz_ctx_t *ctxs[8]; int rc; char err[256]; pthread_t th; int started, joined; } ctxjob_t;
/* device bytes a read set of `bases` bases needs beside the scratch pool when the z-mer index holds every read: the index itself (25 B per z-mer, measured
 * 15.8 B per base), reads (2 bits per base), k-mer seeds + table (< 2.5 B per base), and - transient - the largest of the build temporaries (k-mer index: sort
 * keys + values of 0.19 occurrences per base, double-buffered = 6 B per base, gone before the z-mer index is allocated: BUILD_ZINDEX; z-mer chunks: 8 GB), the arena */
#define WTZ_ZALL_MIN_BASES 2400000000ull
static uint64_t zall_bytes(uint64_t bases){ return (uint64_t)((double)bases * 19.0) + (14ull << 30); }
static void *ctxjob_main(void *arg){
	ctxjob_t *j = (ctxjob_t*)arg;
	for(uint32_t d = 0; d < j->ndev; d++){
		j->rc = wtz_ctx_create(j->devs[d], j->P, j->pool_bytes, &j->ctxs[d]);
		if(j->rc != WTZ_OK){ snprintf(j->err, sizeof j->err, "%s", wtz_last_error()); break; }      /* wtz_last_error is thread-local: keep the text */
	}
	return NULL;
}
typedef struct { wtz_ctx_t *ctx; uint32_t n_rd, K; int rc; char err[256]; int zonly, nozidx; uint32_t zmod, zres; } ixjob_t;
/* the z-mer index of one device for the whole run: every read, or - several parts - the candidate side of its residue class of the indexed reads */
static int zindex_for_part(wtz_ctx_t *ctx, uint32_t n_rd, uint32_t zmod, uint32_t zres){
	if(zmod <= 1) return wtz_zindex_build(ctx);
	uint32_t *m = (uint32_t*)hx_realloc(NULL, 4 * ((size_t)n_rd / zmod + 2)), n = 0;
	for(uint32_t r = zres; r < n_rd; r += zmod) m[n++] = r;
	const int rc = wtz_zindex_build_subset(ctx, m, n);
	free(m); return rc;
}
static void *ixjob_main(void *arg){
	ixjob_t *j = (ixjob_t*)arg; wtz_index_stats_t ist;
	j->rc = j->nozidx ? WTZ_OK : zindex_for_part(j->ctx, j->n_rd, j->zmod, j->zres);
	if(j->rc == WTZ_OK && !j->zonly) j->rc = wtz_index_build(j->ctx, 0, j->n_rd, &j->K, &ist);
	if(j->rc != WTZ_OK){ strncpy(j->err, wtz_last_error(), sizeof j->err - 1); j->err[sizeof j->err - 1] = 0; }
	return NULL;
}

static void *pin_main(void *arg){
	eng_t *E = (eng_t*)arg;
	for(int k = 0; k < 2; k++){
		E->cig_keep[k] = (char*)wtz_host_alloc(E->cig_keep_cap[k] + 1);
		if(!E->cig_keep[k]) E->cig_keep_cap[k] = 0;
	}
	return NULL;
}

/* The z-mer index of a batch (after its candidate rows are known, before its first range).
 *   --zindex-batch (read sets whose all-reads index does not fit: 16 B per base): part d's index is rebuilt from the batch's candidate reads
 *     = d (mod nparts) - with one part: from the batch's queries + all their candidates, one index;
 *   several parts: the batch's queries go into the second, query-side index of EVERY part (wtz_zindex_build_queries).
 * Parts of this process build side by side on their own threads; between ranks the lists travel with WTZ_CMD_ZIDX and every rank builds its own
 * (no reply: a failure shows in the status word of the rank's next reply). */
typedef struct { wtz_ctx_t *ctx; int have_c, have_q; const uint32_t *cl, *ql; uint32_t ncl, nql; int rc; char err[256]; } zjob_t;
static void *zjob_main(void *arg){
	zjob_t *j = (zjob_t*)arg;
	j->rc = zbuild_part(j->ctx, j->have_c, j->cl, j->ncl, j->have_q, j->ql, j->nql);
	if(j->rc != WTZ_OK) snprintf(j->err, sizeof j->err, "%s", wtz_last_error());
	return NULL;
}
static void batch_zindex(eng_t *E, batch_t *b){
	const uint32_t n_all = (uint32_t)(E->st.n_rd + E->st.n_qr), np = b->nparts; const int ranks = g_dist.world > 1;
	const int have_c = E->zbatch > 0, have_q = E->zsplit;
	uint32_t *ql = (uint32_t*)hx_realloc(NULL, 4 * ((size_t)b->nbq + 1)), nql = 0;
	uint32_t *cl[WTZ_DIST_MAX], ncl[WTZ_DIST_MAX]; memset(cl, 0, sizeof cl); memset(ncl, 0, sizeof ncl);
	for(uint32_t s = 0; s < b->nbq; s++) if(b->want[s]) ql[nql++] = b->bq[s];       /* ascending: a batch takes its queries in id order */
	if(have_c){
		uint8_t *mark = (uint8_t*)calloc((size_t)n_all + 1, 1);
		for(uint32_t s = 0; s < b->nbq; s++){
			if(!b->want[s]) continue;
			if(!have_q) mark[b->bq[s]] = 1;          /* one part: queries and candidates share the one index */
			for(uint32_t k = 0; k < b->nrow[s]; k++){ const uint32_t id2 = (uint32_t)(b->rows[(size_t)s * E->stride + k] >> 32); if(id2 < n_all) mark[id2] = 1; }
		}
		for(uint32_t d = 0; d < np; d++) cl[d] = (uint32_t*)hx_realloc(NULL, 4 * ((size_t)n_all / np + 2));
		for(uint32_t r = 0; r < n_all; r++) if(mark[r]){ const uint32_t d = have_q ? DEAL_PART(np, 0, r) : 0; cl[d][ncl[d]++] = r; }
		free(mark);
	}
	const double tz0 = now_s();
	if(ranks){
		wtz_dist_hdr_t h; memset(&h, 0, sizeof h); h.cmd = WTZ_CMD_ZIDX; h.arg[0] = nql; h.arg[1] = (uint64_t)have_c;
		for(uint32_t d = 0; d < np; d++) h.count[d] = ncl[d];
		g_dist.bcast(&h, sizeof h);
		if(nql) g_dist.bcast(ql, 4 * (uint64_t)nql);
		if(have_c) for(uint32_t d = 1; d < np; d++) if(ncl[d]) g_dist.send(cl[d], 4 * (uint64_t)ncl[d], (int)d);
		const int rc = zbuild_part(E->ctx, have_c, cl[0], ncl[0], have_q, ql, nql);
		if(rc != WTZ_OK){ fprintf(stderr, " -- rank 0: the z-mer index of the batch failed: %s --\n", wtz_last_error()); ranks_abort("rank 0 failed"); }
	} else {
		zjob_t zj[8]; pthread_t th[8];
		for(uint32_t d = 0; d < np && d < 8; d++){ zj[d].ctx = b->parts[d].ctx; zj[d].have_c = have_c; zj[d].have_q = have_q; zj[d].cl = cl[d]; zj[d].ncl = ncl[d]; zj[d].ql = ql; zj[d].nql = nql; zj[d].rc = WTZ_OK; }
		for(uint32_t d = 1; d < np; d++) if(pthread_create(&th[d], NULL, zjob_main, &zj[d]) != 0){ fprintf(stderr, " -- cannot start a device thread --\n"); DIE_NOW(); }
		zjob_main(&zj[0]);
		for(uint32_t d = 1; d < np; d++) pthread_join(th[d], NULL);
		for(uint32_t d = 0; d < np; d++) if(zj[d].rc != WTZ_OK){ fprintf(stderr, " -- the z-mer index of the batch failed on device %d: %s --\n", E->devs[d], zj[d].err); DIE_NOW(); }
	}
	pthread_mutex_lock(&E->mu); E->t_gpu += now_s() - tz0; E->t_zbatch += now_s() - tz0; pthread_mutex_unlock(&E->mu);
	for(uint32_t d = 0; d < np; d++) free(cl[d]);
	free(ql);
}

static void cand_scratch_seen(eng_t *E, batch_t *b, uint32_t n_lookup);
static uint32_t cand_batch_cap(const eng_t *E);
/* the next batch of queries (id order) with their candidate heaps (A3) and - where the z-mer index is per batch - its index; 0: no queries left */
static int batch_form(batch_t *b){
	eng_t *E = b->E;
	/* ---- take the next batch of queries (id order) ---- */
	pthread_mutex_lock(&E->mu);
	if(E->cursor >= E->qend){ pthread_mutex_unlock(&E->mu); return 0; }
	const uint32_t B = E->B;
	const int use_pf = b->pf_inflight;
	const uint32_t jend = use_pf ? b->pf_cursor_end : E->qend;      /* a prefetched batch covers exactly the reads the prefetch looked at */
	const uint32_t need = use_pf ? (jend - E->cursor) + 1 : B + 1;  /* saturated reads of the range ride along without candidates */
	if(need > b->capbq){ b->capbq = need; b->bq = (uint32_t*)hx_realloc(b->bq, 4 * (size_t)b->capbq); b->want = (uint8_t*)hx_realloc(b->want, b->capbq); b->ids = (uint32_t*)hx_realloc(b->ids, 4 * (size_t)b->capbq);
		b->rows = (uint64_t*)hx_realloc(b->rows, (size_t)b->capbq * E->stride * 8); b->nrow = (uint32_t*)hx_realloc(b->nrow, 4 * (size_t)b->capbq); }
	b->nbq = 0;
	uint32_t j = E->cursor, nwant = 0;
	for(; j < jend && (use_pf || b->nbq < B); j++){
		if((j % E->n_job) != E->i_job) continue;
		if(E->masked[j]) continue;
		/* reads whose coverage is already saturated (wtzmo.c:808) need no GPU work but stay in the dispatch sequence:
		 * their dispatch is what merges the previous query's masks (one-query masking lag) */
		const int sat = E->rdcovs[j] >= nbest_of(E, j);
		b->want[b->nbq] = (uint8_t)!sat; b->nrow[b->nbq] = 0;
		if(!sat){
			if(E->rows_all){ memcpy(b->rows + (size_t)b->nbq * E->stride, E->rows + (size_t)j * E->stride, (size_t)E->stride * 8); b->nrow[b->nbq] = E->nrow[j]; }
			nwant++;
		}
		b->bq[b->nbq++] = j;
	}
	E->cursor = j;
	b->seq = E->next_seq++; b->holds_turn = 0; b->spec_queries = b->nbq; b->used_queries = 0; b->masked_at_form = E->n_masked;
	E->spec_queries += b->nbq; E->n_batches++;
	{ const uint32_t mb = cand_batch_cap(E); if(E->B < mb) E->B = E->B * 4 > mb ? mb : E->B * 4; else if(E->B > mb) E->B = mb; }      /* ramp-up; corrected at commit */
	pthread_mutex_unlock(&E->mu);
	/* ---- candidate heaps of the batch's queries (A3) ---- */
	if(use_pf){
		const double tg0 = now_s();
		int rc = WTZ_OK, cand_failed = -1;
		if(E->shard) shard_candidates_end(E, b->pf_n, b->pf_rows, b->pf_nr);
		else if(b->nparts == 1){ rc = wtz_candidates_end(b->ctx, b->pf_rows, b->pf_nr); DIE_WTZ(rc, "wtz_candidates_end"); cand_scratch_seen(E, b, b->pf_n); }
		else for(uint32_t d = 0; d < b->nparts; d++){
			part_t *pt = &b->parts[d];
			if(d == 0 && g_dist.world > 1){ wtz_dist_hdr_t h; memset(&h, 0, sizeof h); h.cmd = WTZ_CMD_CAND_END; g_dist.bcast(&h, sizeof h); }
			if(pt->remote){
				const uint64_t sz_rows = (uint64_t)pt->cq_n * E->stride * 8, sz_nr = 4 * (uint64_t)pt->cq_n;
				uint8_t *x = part_xbuf(pt, 8 + sz_rows + sz_nr);
				g_dist.recv(x, 8 + sz_rows + sz_nr, pt->remote);
				uint64_t st = 0; memcpy(&st, x, 8);
				if(st != WTZ_ST_OK){ if(cand_failed < 0) cand_failed = pt->remote; continue; }
				if(pt->cq_n){ memcpy(pt->cq_rows, x + 8, sz_rows); memcpy(pt->cq_nr, x + 8 + sz_rows, sz_nr); }
			} else {
				rc = wtz_candidates_end(pt->ctx, pt->cq_rows, pt->cq_nr);
				if(rc != WTZ_OK && g_dist.world > 1){ fprintf(stderr, " -- rank 0: wtz_candidates_end failed: %s --\n", wtz_last_error()); if(cand_failed < 0) cand_failed = 0; continue; }
				DIE_WTZ(rc, "wtz_candidates_end");
			}
			for(uint32_t k = 0; k < pt->cq_n; k++){ const size_t g = (size_t)k * b->nparts + d; memcpy(b->pf_rows + g * E->stride, pt->cq_rows + (size_t)k * E->stride, (size_t)E->stride * 8); b->pf_nr[g] = pt->cq_nr[k]; }
		}
		if(cand_failed >= 0){ char why[64]; snprintf(why, sizeof why, "rank %d failed in the seed lookup", cand_failed); ranks_abort(why); }
		const double tg1 = now_s();
		b->pf_inflight = 0;
		uint32_t k = 0;
		for(uint32_t s = 0; s < b->nbq; s++){
			if(!b->want[s]) continue;
			while(k < b->pf_n && b->pf_ids[k] < b->bq[s]) k++;
			if(k >= b->pf_n || b->pf_ids[k] != b->bq[s]){ fprintf(stderr, " -- internal error: read %u has no prefetched candidates --\n", b->bq[s]); DIE_NOW(); }
			memcpy(b->rows + (size_t)s * E->stride, b->pf_rows + (size_t)k * E->stride, (size_t)E->stride * 8); b->nrow[s] = b->pf_nr[k];
		}
		pthread_mutex_lock(&E->mu); E->t_gpu += tg1 - tg0; E->t_call[0] += tg1 - tg0; pthread_mutex_unlock(&E->mu);
	} else if(nwant){
		uint32_t n = 0;
		for(uint32_t s = 0; s < b->nbq; s++) if(b->want[s]) b->ids[n++] = b->bq[s];
		uint64_t *rows = (uint64_t*)hx_realloc(NULL, (size_t)n * E->stride * 8); uint32_t *nr = (uint32_t*)hx_realloc(NULL, 4 * (size_t)n);
		n = 0;
		for(uint32_t s = 0; s < b->nbq; s++) if(b->want[s]){ memcpy(rows + (size_t)n * E->stride, b->rows + (size_t)s * E->stride, (size_t)b->nrow[s] * 8); nr[n] = b->nrow[s]; n++; }
		const double tg0 = now_s();
		if(E->shard){ shard_candidates_begin(E, b->ids, n); shard_candidates_end(E, n, rows, nr); }
		else { int rc = wtz_candidates(b->ctx, b->ids, n, rows, nr); DIE_WTZ(rc, "wtz_candidates"); cand_scratch_seen(E, b, n); }
		const double tg1 = now_s();
		n = 0;
		for(uint32_t s = 0; s < b->nbq; s++) if(b->want[s]){ memcpy(b->rows + (size_t)s * E->stride, rows + (size_t)n * E->stride, (size_t)E->stride * 8); b->nrow[s] = nr[n]; n++; }
		free(rows); free(nr);
		pthread_mutex_lock(&E->mu); E->t_gpu += tg1 - tg0; E->t_call[0] += tg1 - tg0; pthread_mutex_unlock(&E->mu);
	}
	if(b->nbq && (E->zbatch > 0 || E->zsplit)) batch_zindex(E, b);
	return 1;
}
/* the seed lookup allocates its scratch per query from the main pool as it goes (tuples of the query: coverage x length) and an exhausted pool ends the run: the
 * batches are capped at what 0.6 of the pool holds at the bytes per query of the requests before (the longest reads come first, so the estimate errs on the safe side) */
static void cand_scratch_seen(eng_t *E, batch_t *b, uint32_t n_lookup){
	wtz_pool_info_t pi;
	if(n_lookup < 32 || b->nparts != 1 || E->shard || b->ctx == NULL || wtz_pool_info(b->ctx, &pi) != WTZ_OK || pi.main_used == 0) return;
	const double bpq = (double)pi.main_used / (double)n_lookup;
	pthread_mutex_lock(&E->mu);
	{ const double keep = E->cand_bpq * 0.7; E->cand_bpq = bpq > keep ? bpq : keep; }      /* follows the measurements down by at most 30 % per request (reads get shorter) */
	E->cand_main_cap = pi.main_cap;
	pthread_mutex_unlock(&E->mu);
}
static uint32_t cand_batch_cap(const eng_t *E){
	uint32_t cap = E->max_batch;
	if(E->cand_bpq > 0 && E->cand_main_cap){ const double c = 0.6 * (double)E->cand_main_cap / E->cand_bpq; if(c < (double)cap) cap = c < 256.0 ? 256u : (uint32_t)c; }
	else if(E->cand_cap0 && cap > E->cand_cap0) cap = E->cand_cap0;
	return cap;
}

static void process_batch(eng_t *E, batch_t *b);
static void *worker_main(void *arg){
	batch_t *b = (batch_t*)arg; eng_t *E = b->E;
	for(;;){
		if(!b->formed && !batch_form(b)) break;       /* formed already: by the batch before it, in front of its last commit */
		b->formed = 0;
		if(b->nbq) process_batch(E, b);
		/* ---- hand the turn to the next batch ---- */
		pthread_mutex_lock(&E->mu);
		while(!b->holds_turn && E->commit_seq != b->seq) pthread_cond_wait(&E->cv, &E->mu);
		E->commit_seq = b->seq + 1;
		if(b->used_queries * 2 < b->spec_queries && E->B > E->first_batch){ E->B /= 2; if(E->B < 8) E->B = 8; }   /* too much discarded work */
		E->last_mask_rate = b->used_queries ? (double)(E->n_masked - b->masked_at_form) / (double)b->used_queries : -1.0;
		pthread_cond_broadcast(&E->cv);
		pthread_mutex_unlock(&E->mu);
		if(b->alt && b->alt->formed) b = b->alt;      /* formed (and started) in front of this batch's last commit */
	}
	return NULL;
}

static double now_s(void){ struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

/* Benchmark hook (bench.py, in-process through libwtzmo_host.so): called with (rep, 0) right before the timed overlap
 * phase of repetition `rep` starts (reads already resident in HBM) and with (rep, 1) after its output is written. */
typedef void (*wtzmo_hook_fn)(int rep, int phase);
static wtzmo_hook_fn g_hook = NULL;
void wtzmo_set_hook(wtzmo_hook_fn f){ g_hook = f; }

#ifdef WTZ_AS_LIB
int wtzmo_main(int argc, char **argv){
	optind = 1;
#else
int main(int argc, char **argv){
#endif
	eng_t *E = (eng_t*)calloc(1, sizeof(eng_t));
	E->st.keep_text = 1;        /* --ingest device */
	E->bpp_decay = getenv("WTZ_BPP_DECAY") ? atof(getenv("WTZ_BPP_DECAY")) : 0.0;      /* experiment: 0 = the estimate only grows (default), e.g. 0.85 = it may fall by 15 % per range */
	wtz_params_c *P = &E->P;
	strlist_t pbs = {0}, flts = {0}, ovls = {0}, obts = {0}, tbas = {0};
	char *output = NULL, *pairoutf = NULL, *statsf = NULL;
	int c, min_rdlen = 0, overwrite = 0, dot_matrix = 0, write_contained = 1, refine = 0, gpu = 0, lib_check = 0, repeat = 1;
	uint64_t pool_gb = 0, pool_mb = 0; float optval; int max_batch_set = 0;
	int n_gpus = 1; const char *gpu_list = NULL;
	/* defaults: wtzmo.c:1543-1588 */
	P->w = 50; P->ew = 800; P->W = 3200; P->M = 2; P->X = -5; P->O = -3; P->E = -1; P->T = -50;
	P->min_score = 200; P->min_id = 0.5f; P->hk = 1; P->hz = 1; P->ksize = 16; P->zsize = 10;
	P->kwin = 800; P->kovl = 300; P->ksave = 4; P->win_rep_norm = 20; P->win_rep_cutoff = 100; P->ncand = 500; P->nbest = 100;
	P->ztot = 300; P->zovl = 200; P->max_kmer_freq = 0; P->max_zmer_freq = 64; P->max_kmer_var = 2;
	P->xvar = 128; P->yvar = 64; P->min_block_len = 160; P->deviation_penalty = 1.0f; P->gap_penalty = 0.05f;
	E->do_align = 1; E->n_idx = 1; E->n_job = 1; E->i_job = 0; E->max_batch = 8192; E->first_batch = 256; E->n_workers = 1;
	pthread_mutex_init(&E->mu, NULL); pthread_cond_init(&E->cv, NULL);
	static struct option lopts[] = { {"stats", required_argument, 0, 1000}, {"gpu", required_argument, 0, 1001}, {"pool-gb", required_argument, 0, 1002},
		{"batch", required_argument, 0, 1003}, {"lib-check", no_argument, 0, 1004}, {"repeat", required_argument, 0, 1005}, {"first-batch", required_argument, 0, 1006}, {"workers", required_argument, 0, 1007}, {"pool-mb", required_argument, 0, 1008}, {"gpus", required_argument, 0, 1010}, {"gpu-list", required_argument, 0, 1011}, {"shard-index", no_argument, 0, 1012}, {"zindex-batch", required_argument, 0, 1013}, {"ingest", required_argument, 0, 1014}, {"binary-out", no_argument, 0, 1015}, {0, 0, 0, 0} };
	while((c = getopt_long(argc, argv, "ht:P:p:Ni:b:J:I:o:9:S:fCH:k:G:z:Z:U:y:d:r:q:l:K:A:B:r:R:L:F:W:w:e:M:X:O:E:T:s:m:nv", lopts, NULL)) != -1){
		switch(c){
			case 1000: statsf = optarg; break;
			case 1001: gpu = atoi(optarg); break;
			case 1002: pool_gb = (uint64_t)atoll(optarg); break;
			case 1003: E->max_batch = (uint32_t)atoi(optarg); if(E->max_batch < 1) E->max_batch = 1; max_batch_set = 1; break;
			case 1004: lib_check = 1; break;
			case 1005: repeat = atoi(optarg); if(repeat < 1) repeat = 1; break;
			case 1006: E->first_batch = (uint32_t)atoi(optarg); if(E->first_batch < 1) E->first_batch = 1; E->first_batch_set = 1; break;
			case 1010: n_gpus = atoi(optarg); if(n_gpus < 1) n_gpus = 1; if(n_gpus > 8) n_gpus = 8; break;
			case 1015: E->binary_out = 1; break;      /* f3: binary record hand-off (include/wtz_ovlb.h) to a consumer that reads it (bin/wtgbo --binary-in, bin/wtovl) */
			case 1014: E->st.keep_text = (strcmp(optarg, "host") != 0); break;      /* f4: `device` (default) = the bases travel as text and are packed to 2 bits on the GPU (wtz_upload_reads_ascii), `host` = packed while reading */
			case 1013: E->zbatch = atoi(optarg) ? 1 : -1; break;      /* 1 = per-batch z-index, 0 = never (default: by the size of the read set) */
			case 1012: E->shard = 1; break;           /* k-mer index sharded over the devices of --gpus / --gpu-list (or over the ranks) */
			case 1011: gpu_list = optarg; break;      /* explicit device ids, e.g. 0,1,2,3 (an id may repeat: two contexts on one GPU, used by the tests) */
			case 1008: pool_mb = (uint64_t)atoll(optarg); break;      /* test hook: a pool small enough to force the batch-splitting path */
			case 1007: E->n_workers = (uint32_t)atoi(optarg); if(E->n_workers < 1) E->n_workers = 1; if(E->n_workers > 8) E->n_workers = 8; break;
			case 'h': return usage();
			case 't': break;
			case 'P': E->n_job = (uint32_t)atoi(optarg); break;
			case 'p': E->i_job = (uint32_t)atoi(optarg); break;
			case 'N': E->do_align = 0; break;
			case 'i': sl_push(&pbs, optarg); break;
			case 'b': sl_push(&obts, optarg); break;
			case 'J': min_rdlen = atoi(optarg); break;
			case 'I': sl_push(&tbas, optarg); break;
			case 'o': output = optarg; break;
			case '9': pairoutf = optarg; break;
			case 'S': { int v = atoi(optarg); if(v < 1) return usage(); P->ksave = (uint32_t)v; } break;
			case 'f': overwrite = 1; break;
			case 'C': write_contained = 0; break;          /* dead switch in the reference: only the side file is suppressed (wtzmo.c:1609,1781) */
			case 'H': { int hk = atoi(optarg); P->hz = (uint32_t)((hk >> 1) & 1); P->hk = (uint32_t)(hk & 1); } break;
			case 'k': P->ksize = (uint32_t)atoi(optarg); break;
			case 'K': P->max_kmer_freq = (uint32_t)atoi(optarg); break;
			case 'z': P->zsize = (uint32_t)atoi(optarg); break;
			case 'Z': P->max_zmer_freq = (uint32_t)atoi(optarg); break;
			case 'U': optval = (float)atof(optarg);
				if(optval < 0){ dot_matrix = 5; break; }
				switch(dot_matrix){
					case 0: P->xvar = (int)optval; break;
					case 1: P->yvar = (int)optval; break;
					case 2: P->min_block_len = (int)optval; break;
					case 3: P->deviation_penalty = optval; break;
					case 4: P->gap_penalty = optval; break;
					default: dot_matrix = 5;
				}
				dot_matrix++;
				break;
			case 'y': P->kwin = (uint32_t)atoi(optarg); break;
			case 'l': P->max_kmer_var = (uint32_t)atoi(optarg); break;
			case 'd': P->kovl = (uint32_t)(int)atof(optarg); break;
			case 'G': E->n_idx = (uint32_t)atoi(optarg); break;
			case 'r': P->ztot = (uint32_t)(int)atof(optarg); break;
			case 'R': P->zovl = (uint32_t)(int)atof(optarg); break;
			case 'q': P->win_rep_cutoff = (float)atoi(optarg); break;
			case 'A': P->ncand = (uint32_t)atoi(optarg); break;
			case 'B': P->nbest = (uint32_t)atoi(optarg); break;
			case 'w': P->w = atoi(optarg); break;
			case 'e': P->ew = atoi(optarg); break;
			case 'W': P->W = atoi(optarg); break;
			case 'M': P->M = atoi(optarg); break;
			case 'X': P->X = atoi(optarg); break;
			case 'O': P->O = atoi(optarg); break;
			case 'E': P->E = atoi(optarg); break;
			case 'T': P->T = atoi(optarg); break;
			case 'L': sl_push(&ovls, optarg); break;
			case 'F': sl_push(&flts, optarg); break;
			case 's': P->min_score = atoi(optarg); break;
			case 'm': P->min_id = (float)atof(optarg); break;
			case 'n': refine = 1; break;
			case 'v': break;
			default: return usage();
		}
	}
	E->keep_order = pairoutf != NULL;
	if(lib_check){ printf("libwtzmo_hip: %d device(s)\n", wtz_device_count()); return wtz_device_count() > 0 ? 0 : 3; }
	if(output == NULL) return usage();
	if(!overwrite && strcmp(output, "-") && access(output, F_OK) == 0){ fprintf(stderr, "File exists! '%s'\n\n", output); return usage(); }
	if(pbs.n == 0) return usage();
	if(P->ksize > 32 || P->ksize < 5) return usage();
	if(P->zsize > 16 || P->zsize < 5) return usage();
	P->refine = refine;
	if(E->n_idx < 1) E->n_idx = 1;
	if(E->n_job < 1) E->n_job = 1;
	P->max_overhang = 2 * P->xvar; P->kstep = P->kwin / 2; P->dot_matrix = dot_matrix;

	/* devices: --gpu <id> (one), --gpus N (ids 0..N-1) or --gpu-list; every device gets the reads and builds both indexes (replicated) */
	E->ndev = 0;
	if(gpu_list){ const char *q = gpu_list; while(*q && E->ndev < 8){ E->devs[E->ndev++] = atoi(q); while(*q && *q != ',') q++; if(*q == ',') q++; } }
	else if(n_gpus > 1){ for(int d = 0; d < n_gpus; d++) E->devs[E->ndev++] = d; }
	if(E->ndev == 0){ E->devs[0] = gpu; E->ndev = 1; }
	if(E->ndev > 1 && (E->n_idx > 1 || E->n_workers > 1)){
		fprintf(stderr, "[wtzmo-mi355x] -G / --workers run on one device: --gpus ignored\n"); E->ndev = 1;
	}
	static ctxjob_t cj; memset(&cj, 0, sizeof cj);
	/* Large inputs on ONE device (configs[3]: 10 Gbp): the all-reads z-mer index (16 B per base, built once, in chunks) beats the per-batch one (rebuilt for every
	 * batch's queries + candidates: 8.1 of 28.8 s per configs[3]-shape step) whenever it fits BESIDE the scratch pool - so the pool, which is allocated now, on a
	 * helper thread, while the reads are still being parsed, is sized from the input files' sizes (a byte of FASTA is at most a base; gz: x4): what the indexes
	 * will need stays free.  The exact test follows once the reads are counted (below); a pipe or an underestimate just means the per-batch form as before. */
	uint64_t pool_auto = 0;
	if(!pool_gb && !pool_mb && E->ndev == 1 && g_dist.world == 1 && E->n_workers == 1 && E->zbatch <= 0 && !getenv("WTZ_NO_ZALL")){
		uint64_t est = 0; int known = 1;
		for(int k = 0; k < pbs.n + tbas.n; k++){
			const char *fn = k < pbs.n ? pbs.a[k] : tbas.a[k - pbs.n]; struct stat sb;
			if(stat(fn, &sb) != 0 || !S_ISREG(sb.st_mode)){ known = 0; break; }
			const size_t ln = strlen(fn);
			est += (ln > 3 && !strcmp(fn + ln - 3, ".gz")) ? (uint64_t)sb.st_size * 4 : (uint64_t)sb.st_size;
		}
		uint64_t fr_b = 0, tot_b = 0;
		if(known && est > WTZ_ZALL_MIN_BASES && wtz_device_memory(E->devs[0], &fr_b, &tot_b) == WTZ_OK){
			const uint64_t need = zall_bytes(est);
			if(fr_b > need + (56ull << 30)){ pool_auto = fr_b - need - (8ull << 30); if(pool_auto > (128ull << 30)) pool_auto = 128ull << 30; }      /* 8 GB: the arena, and the estimate against the exact count */
		}
	}
	cj.P = P; cj.pool_bytes = pool_mb ? pool_mb << 20 : (pool_gb ? pool_gb << 30 : pool_auto); cj.ndev = E->ndev; for(uint32_t d = 0; d < E->ndev; d++) cj.devs[d] = E->devs[d];
	cj.started = (pthread_create(&cj.th, NULL, ctxjob_main, &cj) == 0);      /* from here to the join every error path leaves through DIE_NOW (_exit): exit() would tear HIP down under that thread */

	/* ---- load reads (wtzmo.c:1691-1729) ---- */
	hx_str_t name = {0}, seq = {0};
	hx_reader_t *fr = hx_reader_open(pbs.a, pbs.n);
	if(fr == NULL){ fprintf(stderr, " -- Cannot open %s --\n", pbs.a[0]); DIE_NOW(); }
	fprintf(stderr, "[wtzmo-mi355x] loading long reads\n");
	while(hx_reader_seq(fr, &name, &seq)){
		if((int)seq.n < min_rdlen) continue;
		hx_store_add(&E->st, name.s ? name.s : "", name.n, seq.s ? seq.s : "", seq.n);
		E->st.n_rd++;
	}
	hx_reader_close(fr);
	hx_sort_exact(E->st.reads, E->st.n_rd, sizeof(hx_read_t), gt_read, NULL);
	if(tbas.n){
		if((fr = hx_reader_open(tbas.a, tbas.n)) == NULL) DIE_NOW();
		while(hx_reader_seq(fr, &name, &seq)){
			if((int)seq.n < min_rdlen) continue;
			hx_store_add(&E->st, name.s ? name.s : "", name.n, seq.s ? seq.s : "", seq.n);
			E->st.n_qr++;
		}
		hx_reader_close(fr);
	}
	const uint32_t n_rd = E->st.n_rd, n_all = E->st.n_all;
	fprintf(stderr, "[wtzmo-mi355x] %u reads (+%u query-only), %llu bp\n", n_rd, E->st.n_qr, (unsigned long long)E->st.nbase);
	E->masked = (uint8_t*)calloc((size_t)n_all + 1, 1);
	E->rdcovs = (uint32_t*)calloc((size_t)n_all + 1, 4);
	E->closed_touch = (uint32_t*)calloc((size_t)n_all + 1, 4);
	g_pend_eng = E;
	hx_names_t nm; hx_names_build(&nm, E->st.reads, n_rd);
	char *cols[4];
	if(obts.n){
		if((fr = hx_reader_open(obts.a, obts.n)) == NULL) DIE_NOW();
		while(hx_reader_line(fr) != -1){
			if(fr->line[0] == '#') continue;
			int nc = 0; char *p = fr->line;
			while(nc < 4){ cols[nc++] = p; while(*p && *p != '\t' && *p != ' ') p++; if(!*p) break; *p++ = 0; }
			if(nc < 3) continue;
			uint32_t id = hx_names_get(&nm, cols[0]); int coff = atoi(cols[1]), clen = atoi(cols[2]);
			if(id == 0xFFFFFFFFu) continue;
			hx_read_t *rd = &E->st.reads[id];
			if(coff < 0 || coff + clen > (int)rd->len) continue;
			rd->off += (uint64_t)coff; rd->len = (uint32_t)clen;
		}
		hx_reader_close(fr);
	}
	if(flts.n){
		if((fr = hx_reader_open(flts.a, flts.n)) == NULL) DIE_NOW();
		while(hx_reader_line(fr) != -1){
			if(fr->line[0] == '#') continue;
			uint32_t id = hx_names_get(&nm, fr->line);
			if(id != 0xFFFFFFFFu) E->masked[id] = 1;
		}
		hx_reader_close(fr);
	}
	if(ovls.n){
		if((fr = hx_reader_open(ovls.a, ovls.n)) == NULL) DIE_NOW();
		while(hx_reader_line(fr) != -1){
			if(fr->line[0] == '#') continue;
			int nc = 0; char *p = fr->line;
			while(nc < 4){ cols[nc++] = p; while(*p && *p != '\t' && *p != ' ') p++; if(!*p) break; *p++ = 0; }
			if(nc < 2) continue;
			uint32_t a = hx_names_get(&nm, cols[0]), b = hx_names_get(&nm, cols[1]);
			if(a == 0xFFFFFFFFu || b == 0xFFFFFFFFu) continue;
			if(hx_set_put(&E->closed, hx_pair_key(a, b))) order_push(E, hx_pair_key(a, b));
			else order_push(E, hx_pair_key(a, b) | 1u);        /* a repeated -L line is still a put_u64hash of the reference (wtzmo.c:1769): bit 0 marks it for the -9 replay */
		}
		hx_reader_close(fr);
	}
	/* ---- device ---- */
	const uint64_t pool_bytes = pool_mb ? pool_mb << 20 : pool_gb << 30;
	E->rdlen = (uint32_t*)hx_realloc(NULL, 4 * ((size_t)n_all + 1));
	uint64_t *rdoff = (uint64_t*)hx_realloc(NULL, 8 * ((size_t)n_all + 1));
	{ uint64_t tot = 0; uint32_t nq = E->st.n_qr ? E->st.n_qr : n_rd, b0 = E->st.n_qr ? n_rd : 0;
	  for(uint32_t i = 0; i < n_all; i++){ E->rdlen[i] = E->st.reads[i].len; rdoff[i] = E->st.reads[i].off; }
	  for(uint32_t i = 0; i < nq; i++) tot += E->rdlen[b0 + i];
	  E->avg_rdlen = nq ? (uint32_t)(tot / nq) : 10000; if(E->avg_rdlen == 0) E->avg_rdlen = 1; }        /* wtzmo.c:361-368 */
	if(g_dist.world > 1){
		if(g_dist.world > WTZ_DIST_MAX || !g_dist.bcast || !g_dist.send || !g_dist.recv){ fprintf(stderr, " -- wtzmo_set_dist: bad rank setup --\n"); DIE_NOW(); }
		if(E->n_idx > 1 || E->n_workers > 1 || E->ndev > 1 || E->n_job > 1){ fprintf(stderr, " -- ranks (one process per GPU) exclude -G, -P, --workers and --gpus --\n"); DIE_NOW(); }
	}
	E->zsplit = (g_dist.world > 1 || E->ndev > 1) && !getenv("WTZ_NO_ZSPLIT");
	{ /* all-reads z-index: 16 B per base - with several parts only 1 / nparts of it per device, so the per-batch form starts nparts times later */
	  const uint64_t zparts = E->zsplit ? (g_dist.world > 1 ? (uint64_t)g_dist.world : E->ndev) : 1;
	  int zall = 0;
	  if(E->zbatch == 0 && E->st.nbase > WTZ_ZALL_MIN_BASES * zparts && E->n_workers == 1 && zparts == 1 && !getenv("WTZ_NO_ZALL")){
		/* the contexts (scratch pools) exist by now: does the all-reads index fit into what they left? */
		if(cj.started){ pthread_join(cj.th, NULL); cj.started = 0; cj.joined = 1; }
		uint64_t fr_b = 0, tot_b = 0;
		if(cj.joined && cj.rc == WTZ_OK && wtz_device_memory(E->devs[0], &fr_b, &tot_b) == WTZ_OK && fr_b > zall_bytes(E->st.nbase)){
			zall = 1;
			/* the seed lookup's scratch grows with the tuples of a query (coverage x length: ~450 k at the configs[3] shape = 26 MB per query) and the pool is the smaller one
			 * chosen above: batches of 1 024 queries until the first request has been measured, then what 0.6 of the pool holds (round 5: 2 048 queries had asked for 54.4 GB of a 49 GB main pool before the group list took the place of the dead tuple list) */
			if(!max_batch_set) E->cand_cap0 = 1024;      /* until the first request has been measured (cand_batch_cap) */
			fprintf(stderr, "[wtzmo-mi355x] %llu read bases: all-reads z-mer index (%.0f GB of %.0f GB free beside the scratch pool)\n", (unsigned long long)E->st.nbase, zall_bytes(E->st.nbase) / 1e9, fr_b / 1e9);
		}
	  }
	  if(!zall && E->zbatch == 0 && E->st.nbase > WTZ_ZALL_MIN_BASES * zparts && E->n_workers == 1){ E->zbatch = 1; fprintf(stderr, "[wtzmo-mi355x] %llu read bases: the z-mer index is built per batch of queries (--zindex-batch 0 to force the all-reads index)\n", (unsigned long long)E->st.nbase); } }
	if(E->zbatch > 0 && E->n_workers > 1){ fprintf(stderr, " -- --zindex-batch excludes --workers --\n"); DIE_NOW(); }
	{ const uint32_t zb_max = getenv("WTZ_ZBATCH_MAX") ? (uint32_t)atoi(getenv("WTZ_ZBATCH_MAX")) : 1024u;      /* queries per batch when the z-mer index is rebuilt per batch: bounds its size (queries + <= -A candidates each); configs[3]-shape whole job: 36.0 s with 512, 33.3 s with 1 024, 33.1 s with 2 048 */
	  if(E->zbatch > 0 && E->max_batch > zb_max) E->max_batch = zb_max; }
	if(E->shard && (E->n_idx > 1 || E->n_workers > 1)){ fprintf(stderr, " -- --shard-index excludes -G and --workers --\n"); DIE_NOW(); }
	int rc;
	if(cj.started) pthread_join(cj.th, NULL); else if(!cj.joined) ctxjob_main(&cj);
	if(cj.rc != WTZ_OK){ fprintf(stderr, " -- wtz_ctx_create failed: %s --\n", cj.err); DIE_NOW(); }
	for(uint32_t d = 0; d < E->ndev; d++){
		E->ctxs[d] = cj.ctxs[d];
		if(E->st.keep_text && E->st.bits == NULL){
			/* f4: seq2basebank on the device (dna.h:397-410); the packed bank comes back once for the other contexts of this process */
			const double ti0 = now_s(); uint64_t n_other = 0;
			rc = wtz_upload_reads_ascii(E->ctxs[d], E->st.text, E->st.nbase, rdoff, E->rdlen, n_all, 0, &n_other); DIE_WTZ(rc, "wtz_upload_reads_ascii");
			const uint64_t nw = (E->st.nbase + 31) / 32;
			E->st.bits = (uint64_t*)hx_realloc(NULL, 8 * (nw + 2)); E->st.capw = nw + 2; E->st.bits[nw] = E->st.bits[nw + 1] = 0;
			rc = wtz_fetch_read_bits(E->ctxs[d], E->st.bits, nw); DIE_WTZ(rc, "wtz_fetch_read_bits");
			free(E->st.text); E->st.text = NULL; E->st.captext = 0;
			{ wtz_counters_t ic; if(wtz_get_counters(E->ctxs[d], &ic) == WTZ_OK){ E->ing_ms = ic.ms_ingest; E->ing_bytes = ic.bytes_ingest_algo; } }
			{ wtz_counters_t ic; if(wtz_get_counters(E->ctxs[d], &ic) == WTZ_OK) fprintf(stderr, "[wtzmo-mi355x] %llu bases packed on the device (%llu not ACGT): kernels %.2f ms = %.0f GB/s, with the copies %.0f ms\n",
				(unsigned long long)E->st.nbase, (unsigned long long)n_other, ic.ms_ingest, ic.ms_ingest > 0 ? (double)ic.bytes_ingest_algo / ic.ms_ingest / 1e6 : 0.0, 1e3 * (now_s() - ti0)); }
			continue;
		}
		rc = wtz_upload_reads(E->ctxs[d], E->st.bits, (E->st.nbase + 31) / 32, rdoff, E->rdlen, n_all); DIE_WTZ(rc, "wtz_upload_reads");
	}
	E->ctx = E->ctxs[0];
	{ wtz_pool_info_t pi; if(wtz_pool_info(E->ctx, &pi) == WTZ_OK) E->main_cap = pi.main_cap * E->ndev; }
	/* page-lock the first worker's CIGAR text buffer while the indexes are built (pinning ~100 MB takes about as long as they do) */
	pthread_t pin_th; int pin_started = 0;
	if(E->do_align && E->cig_keep[0] == NULL){
		uint64_t cap = E->st.nbase / 4 * 3; if(cap < ((uint64_t)16 << 20)) cap = (uint64_t)16 << 20; if(cap > ((uint64_t)256 << 20)) cap = (uint64_t)256 << 20;
		E->cig_keep_cap[0] = E->cig_keep_cap[1] = cap;
		pin_started = (pthread_create(&pin_th, NULL, pin_main, E) == 0);
		if(!pin_started) E->cig_keep_cap[0] = E->cig_keep_cap[1] = 0;
	}
	E->out = strcmp(output, "-") ? fopen(output, "w") : stdout;
	if(E->out == NULL){ fprintf(stderr, " -- Cannot write %s --\n", output); exit(1); }
	setvbuf(E->out, NULL, _IOFBF, 8u << 20);
	E->stride = P->ncand + 1;
	E->pend.rd_id = 0xFFFFFFFFu;
	/* --repeat: run the whole overlap phase several times on the reads already resident in HBM (benchmarking) */
	uint8_t *masked0 = (uint8_t*)hx_realloc(NULL, (size_t)n_all + 1); memcpy(masked0, E->masked, (size_t)n_all + 1);
	const size_t n_order0 = E->n_order;      /* the -L preloads */
	size_t nclosed0 = 0; uint64_t *closed0 = (uint64_t*)hx_realloc(NULL, 8 * (E->closed.n + 1));
	for(size_t i = 0; i < E->closed.cap; i++) if(E->closed.tab[i] != ~0ULL) closed0[nclosed0++] = E->closed.tab[i];
	if(statsf){ FILE *sf = fopen(statsf, "w"); if(sf) fclose(sf); }
	stale_job_t stale_job; memset(&stale_job, 0, sizeof stale_job);
	stale_job.path = (char*)hx_realloc(NULL, strlen(output) + 16); sprintf(stale_job.path, "%s.prev", output);
	for(int rep = 0; rep < repeat; rep++){
		if(rep){
			memcpy(E->masked, masked0, (size_t)n_all + 1); memset(E->rdcovs, 0, 4 * ((size_t)n_all + 1));
			free(E->closed.tab); memset(&E->closed, 0, sizeof E->closed);
			for(size_t i = 0; i < nclosed0; i++) hx_set_put(&E->closed, closed0[i]);
			E->n_order = n_order0;
			E->pair_bp = E->n_pairs = E->nrec = 0;
			memset(E->t_cq, 0, sizeof E->t_cq); memset(E->t_cq1, 0, sizeof E->t_cq1); E->n_spec_hit = E->n_spec_miss = 0; E->t_spec_wait = E->t_spec_ctl = E->t_flush = 0; memset(E->closed_touch, 0, 4 * ((size_t)n_all + 1)); E->t_gpu = E->t_commit = E->t_zbatch = 0; memset(E->t_call, 0, sizeof E->t_call); E->t_io[0] = E->t_io[1] = 0; E->spec_pairs = E->used_pairs = E->spec_items = E->used_items = E->spec_queries = E->used_queries = 0;
			E->rows_all = 0; E->n_batches = 0; E->n_split = 0; E->n_ranges = 0; E->bytes_per_pair = 0;      /* every repeat plans like a cold run: probe range first */
			E->pend.rd_id = 0xFFFFFFFFu; E->pend.nhit = E->pend.nmask = E->pend.nclosed = E->pend.nseed = 0;
			if(strcmp(output, "-")){
				/* the previous repeat's file (3 GB at configs[2]) is neither truncated in place nor removed inline: freeing its pages took
				 * ~0.4 s of the next repeat's timed region, and on a side thread BESIDE THE INDEX BUILDS it slowed their device allocations
				 * (z-index 140 -> 800 ms).  A regular file is renamed to <output>.prev and removed by a side thread once this repeat's
				 * index builds are done (stale_start below): at most ONE stale file exists at any time, whatever --repeat says.  Anything that
				 * is not a regular file (/dev/null, a FIFO, a symlink) is never renamed: it is simply opened again. */
				stale_join(&stale_job);
				struct stat sb;
				if(lstat(output, &sb) == 0 && S_ISREG(sb.st_mode)){
					unlink(stale_job.path);
					if(rename(output, stale_job.path) == 0) stale_job.pending = 1;
				}
				E->out = fopen(output, "w"); if(E->out == NULL){ fprintf(stderr, " -- Cannot write %s: %s --\n", output, strerror(errno)); DIE_NOW(); } setvbuf(E->out, NULL, _IOFBF, 8u << 20);
			}
			wtz_reset_counters(E->ctx);
		}
		if(E->binary_out && g_dist.rank == 0){      /* the name table: ids in the records are this run's read ids */
			const char **nmv = (const char**)hx_realloc(NULL, sizeof(char*) * ((size_t)n_all + 1));
			for(uint32_t i = 0; i < n_all; i++) nmv[i] = E->st.reads[i].name;
			if(wtz_ovlb_write_header(E->out, n_all, nmv, E->rdlen) != 0){ fprintf(stderr, " -- cannot write the binary overlap header --\n"); DIE_NOW(); }
			free(nmv);
		}
		out_start(E->out, E->st.reads, E->binary_out);
		if(g_hook) g_hook(rep, 0);
		const double t0 = now_s();
		pthread_t ixth[8]; ixjob_t ixj[8];
		const uint32_t zmod = E->zsplit ? (g_dist.world > 1 ? (uint32_t)g_dist.world : E->ndev) : 1u;
		for(uint32_t d = 1; d < E->ndev; d++){ ixj[d].ctx = E->ctxs[d]; ixj[d].n_rd = n_rd; ixj[d].K = P->max_kmer_freq; ixj[d].zonly = E->shard; ixj[d].nozidx = E->zbatch > 0; ixj[d].zmod = zmod; ixj[d].zres = d; if(pthread_create(&ixth[d], NULL, ixjob_main, &ixj[d]) != 0) DIE_NOW(); }
		/* the z-mer index of this context is built AFTER its (last) k-mer index: only the pair stages read it, and the k-mer build's sort buffers (6 B per base) are
		 * gone by then - on a 10 Gbp read set the two do not fit side by side next to the scratch pool */
#define BUILD_ZINDEX() do { if(E->zbatch <= 0){ rc = zindex_for_part(E->ctx, n_rd, zmod, g_dist.world > 1 ? (uint32_t)g_dist.rank : 0u); DIE_WTZ(rc, "wtz_zindex_build"); } } while(0)
		/* ---- index parts (-G, wtzmo.c:1276-1303) ---- */
		uint32_t pbbeg = 0, pbend = 0, K = P->max_kmer_freq;
		wtz_index_stats_t ist;
		uint32_t *ids = (uint32_t*)hx_realloc(NULL, 4 * ((size_t)n_all + 1));
		if(E->n_idx > 1){ E->rows_all = 1; E->rows = (uint64_t*)hx_realloc(E->rows, (size_t)n_all * E->stride * 8); E->nrow = (uint32_t*)hx_realloc(E->nrow, (size_t)n_all * 4); memset(E->nrow, 0, (size_t)n_all * 4); }
		double t_index = 0;
		if(E->shard){
			/* the other devices' z-index threads use their contexts: join them first */
			for(uint32_t d = 1; d < E->ndev; d++){ pthread_join(ixth[d], NULL); if(ixj[d].rc != WTZ_OK){ fprintf(stderr, " -- index build on device %d failed: %s --\n", E->devs[d], ixj[d].err); DIE_NOW(); } }
			const double ti = now_s();
			shard_index_build(E, n_rd, &K);
			t_index += now_s() - ti;
			BUILD_ZINDEX();
		}
		for(uint32_t i_idx = 0; i_idx < E->n_idx && !E->shard; i_idx++){
			pbbeg = pbend; pbend = pbbeg + (n_rd + E->n_idx - 1) / E->n_idx;
			const double ti = now_s();
			rc = wtz_index_build(E->ctx, pbbeg, pbend > n_rd ? n_rd : pbend, &K, &ist); DIE_WTZ(rc, "wtz_index_build");
			t_index += now_s() - ti;
			fprintf(stderr, "[wtzmo-mi355x] index %u/%u: %llu k-mer occurrences, %llu distinct, %llu kept, cutoff %u\n", i_idx + 1, E->n_idx,
				(unsigned long long)ist.n_occ, (unsigned long long)ist.n_distinct, (unsigned long long)ist.n_kept, K);
			if(i_idx + 1 >= E->n_idx){ BUILD_ZINDEX(); break; }
			/* just_query pass: accumulate candidate heaps of every unmasked read of this job */
			uint32_t n = 0;
			for(uint32_t j = 0; j < n_rd; j++){ if((j % E->n_job) != E->i_job) continue; if(E->masked[j]) continue; ids[n++] = j; }
			for(uint32_t a = 0; a < n; a += 4096){
				uint32_t m = n - a < 4096 ? n - a : 4096;
				uint64_t *tmp_rows = (uint64_t*)hx_realloc(NULL, (size_t)m * E->stride * 8); uint32_t *tmp_n = (uint32_t*)hx_realloc(NULL, (size_t)m * 4);
				for(uint32_t k = 0; k < m; k++){ memcpy(tmp_rows + (size_t)k * E->stride, E->rows + (size_t)ids[a + k] * E->stride, (size_t)E->stride * 8); tmp_n[k] = E->nrow[ids[a + k]]; }
				rc = wtz_candidates(E->ctx, ids + a, m, tmp_rows, tmp_n); DIE_WTZ(rc, "wtz_candidates");
				for(uint32_t k = 0; k < m; k++){
					/* closed filter + exact sort + trim are applied in these passes too (wtzmo.c:813-823) */
					uint32_t nc = tmp_n[k]; cand_t *cd = (cand_t*)hx_realloc(NULL, sizeof(cand_t) * (nc + 1));
					for(uint32_t x = 0; x < nc; x++){ cd[x].e = tmp_rows[(size_t)k * E->stride + x]; cd[x].pidx = 0; cd[x].pad = 0;
						if(hx_set_has(&E->closed, hx_pair_key(ids[a + k], cd[x].e >> 32))) cd[x].e &= 0xFFFFFFFF00000000ULL; }
					sort_cands_exact(cd, nc);
					while(nc && (uint32_t)cd[nc - 1].e == 0) nc--;
					for(uint32_t x = 0; x < nc; x++) E->rows[(size_t)ids[a + k] * E->stride + x] = cd[x].e;
					E->nrow[ids[a + k]] = nc; free(cd);
				}
				free(tmp_rows); free(tmp_n);
			}
		}
		for(uint32_t d = 1; d < E->ndev && !E->shard; d++){ pthread_join(ixth[d], NULL); if(ixj[d].rc != WTZ_OK){ fprintf(stderr, " -- index build on device %d failed: %s --\n", E->devs[d], ixj[d].err); DIE_NOW(); } }
		stale_start(&stale_job);       /* index builds (and their device allocations) are done: drop the previous repeat's file in the background */
		/* ---- queries: pipelined batches on n_workers contexts (own stream + pool each, indexes shared) ---- */
		if(pin_started){ pthread_join(pin_th, NULL); pin_started = 0; }
		{
			const uint32_t qbeg = E->st.n_qr ? n_rd : 0;
			E->qend = E->st.n_qr ? n_rd + E->st.n_qr : n_rd;
			E->cursor = qbeg; E->B = E->max_batch < E->first_batch ? E->max_batch : E->first_batch;
			if(!E->first_batch_set && E->n_workers == 1){
				/* a job with few queries (a stripe of -P N on a small input) runs them as ONE batch: every batch costs a fixed set of launch
				 * tails (the longest extension of each launch), which outweighs the extra speculation of a large first batch
				 * (measured on the E. coli shape with -P 8: 0.186 -> 0.165 s; any batch size gives the same output) */
				uint32_t nq = 0; for(uint32_t j = qbeg; j < E->qend; j++) if((j % E->n_job) == E->i_job) nq++;
				if(nq <= E->max_batch) E->B = E->max_batch;
			}
			E->next_seq = 0; E->commit_seq = 0; E->last_mask_rate = -1.0; E->n_masked = 0; E->cand_bpq = 0; E->pair_row_ratio = 1.0;
			uint32_t nw = E->rows_all ? 1 : E->n_workers;          /* -G keeps per-read heaps that the commit rewrites: one batch at a time */
			/* one worker, one process: a SECOND batch on the same context(s), formed and started in front of the first one's last commit (process_batch);
			 * WTZ_BATCH_OVERLAP=0 / WTZ_RANGE_OVERLAP=0 keep one batch at a time */
			const int nalt = (nw == 1 && g_dist.world == 1 && !E->rows_all && !(getenv("WTZ_BATCH_OVERLAP") && !atoi(getenv("WTZ_BATCH_OVERLAP"))) && !(getenv("WTZ_RANGE_OVERLAP") && !atoi(getenv("WTZ_RANGE_OVERLAP")))) ? 1 : 0;
			const uint32_t nw_run = nw; nw += (uint32_t)nalt;
			batch_t *bs = (batch_t*)calloc(nw, sizeof(batch_t)); pthread_t *th = (pthread_t*)calloc(nw, sizeof(pthread_t));
			const uint32_t nparts = g_dist.world > 1 ? (uint32_t)g_dist.world : (nw_run == 1 ? E->ndev : 1);
			for(uint32_t w = 0; w < nw; w++){
				bs[w].E = E;
				bs[w].nparts = nparts; bs[w].parts = (part_t*)calloc(nparts, sizeof(part_t)); bs[w].spare = (part_t*)calloc(nparts, sizeof(part_t)); bs[w].cparts = bs[w].parts;
				for(uint32_t d = 0; d < nparts; d++){
					part_t *pt = &bs[w].parts[d]; const uint32_t slot = w * nparts + d;
					pt->ext_base = slot < 8 ? (int)slot * 2 : -1;
					if(slot < 8) for(int k = 0; k < 2; k++){ pt->cigs[k] = E->cig_keep[slot * 2 + k]; pt->capcigs[k] = E->cig_keep_cap[slot * 2 + k]; E->cig_keep[slot * 2 + k] = NULL; E->cig_keep_cap[slot * 2 + k] = 0; }
					if(g_dist.world > 1){ pt->remote = (int)d; pt->ctx = d == 0 ? E->ctx : NULL; }       /* part r belongs to rank r */
					else if(w == 0 || (nalt && w == 1)) pt->ctx = E->ctxs[d];       /* the alternate batch runs on the same contexts, never at the same time */
					else { rc = wtz_ctx_clone(E->ctx, pool_bytes, &pt->ctx); DIE_WTZ(rc, "wtz_ctx_clone"); }
				}
				bs[w].ctx = bs[w].parts[0].ctx;
			}
			if(nalt){ bs[0].alt = &bs[1]; bs[1].alt = &bs[0]; bs[1].shares_ctx = 1; }
			if(g_dist.rank > 0){
				/* this rank serves rank 0's requests with its GPU; plan, commit and output are rank 0's */
				part_t *me = &bs[0].parts[0]; me->ctx = E->ctx; me->remote = 0; me->ext_base = -1;
				remote_loop(E, me);
			} else {
				for(uint32_t w = 1; w < nw_run; w++) pthread_create(&th[w], NULL, worker_main, &bs[w]);
				worker_main(&bs[0]);
				for(uint32_t w = 1; w < nw_run; w++) pthread_join(th[w], NULL);
				if(g_dist.world > 1){ wtz_dist_hdr_t h; memset(&h, 0, sizeof h); h.cmd = WTZ_CMD_DONE; g_dist.bcast(&h, sizeof h); }
			}
			for(uint32_t w = 0; w < nw; w++){
				for(uint32_t d = 0; d < nparts; d++){
					part_t *pt = &bs[w].parts[d]; const uint32_t slot = w * nparts + d;
					if((w || d) && pt->ctx && !bs[w].shares_ctx){      /* counters of the other contexts: work adds up, kernel times of parallel devices do not (the longest counts) */
						wtz_counters_t cw; wtz_get_counters(pt->ctx, &cw);
						if(w){ E->extra_ms[0] += cw.ms_candidates; E->extra_ms[1] += cw.ms_pairs; E->extra_ms[2] += cw.ms_winalign; E->extra_ms[3] += cw.ms_stitch; E->extra_ms[4] += cw.ms_ext; E->extra_ms[5] += cw.ms_gap; }
						E->extra_u64[0] += cw.cells_shift; E->extra_u64[1] += cw.cells_fixed; E->extra_u64[2] += cw.cells_global; E->extra_u64[3] += cw.bytes_seed_algo; E->extra_u64[4] += cw.n_extjobs; E->extra_u64[6] += cw.bytes_zmer_algo;
						if(cw.pool_peak > E->extra_u64[5]) E->extra_u64[5] = cw.pool_peak;
						if(w) wtz_ctx_destroy(pt->ctx); else wtz_reset_counters(pt->ctx);
					}
					free(pt->cq_ids); free(pt->cq_nr); free(pt->cq_rows);
					free(pt->pq); free(pt->pc); free(pt->sum); free(pt->box_off); free(pt->boxes); free(pt->item_of); free(pt->it_pair); free(pt->it_dir); free(pt->aln);
					for(int k = 0; k < 2; k++){ if(slot < 8){ E->cig_keep[slot * 2 + k] = pt->cigs[k]; E->cig_keep_cap[slot * 2 + k] = pt->capcigs[k]; } else wtz_host_free(pt->cigs[k]); }
				}
				for(uint32_t d = 0; d < nparts; d++){ part_t *sp = &bs[w].spare[d]; free(sp->sum); free(sp->box_off); free(sp->boxes); free(sp->item_of); free(sp->it_pair); free(sp->it_dir); free(sp->aln); }
				free(bs[w].spare);
				free(bs[w].parts); free(bs[w].start_job);
				free(bs[w].bq); free(bs[w].want); free(bs[w].ids); free(bs[w].rows); free(bs[w].nrow); free(bs[w].rowpair); free(bs[w].pf_ids); free(bs[w].pf_rows); free(bs[w].pf_nr);
			}
			free(bs); free(th);
		}
		flush_pending(E);
		{ const double tw0 = now_s(); out_finish(); E->t_io[1] += now_s() - tw0; }
		const double t1 = now_s();
		if(strcmp(output, "-")) fclose(E->out); else fflush(stdout);
		if(g_hook) g_hook(rep, 1);
		wtz_counters_t cn; wtz_get_counters(E->ctx, &cn);
		cn.ms_candidates += E->extra_ms[0]; cn.ms_pairs += E->extra_ms[1]; cn.ms_winalign += E->extra_ms[2]; cn.ms_stitch += E->extra_ms[3]; cn.ms_ext += E->extra_ms[4]; cn.ms_gap += E->extra_ms[5];
		cn.cells_shift += E->extra_u64[0]; cn.cells_fixed += E->extra_u64[1]; cn.cells_global += E->extra_u64[2]; cn.bytes_seed_algo += E->extra_u64[3]; cn.n_extjobs += E->extra_u64[4]; cn.bytes_zmer_algo += E->extra_u64[6];
		if(E->extra_u64[5] > cn.pool_peak) cn.pool_peak = E->extra_u64[5];
		memset(E->extra_ms, 0, sizeof E->extra_ms); memset(E->extra_u64, 0, sizeof E->extra_u64);
		fprintf(stderr, "[wtzmo-mi355x] %llu records, %llu pairs aligned, %llu pair-bp, %.3f s (index %.3f s)\n", (unsigned long long)E->nrec, (unsigned long long)E->n_pairs, (unsigned long long)E->pair_bp, t1 - t0, t_index);
		fprintf(stderr, "[wtzmo-mi355x] host seconds: in GPU-stage calls %.3f, commit %.3f, per-batch z-index %.3f; writer thread: formatting %.3f, write %.3f; waiting for it: %.3f before buffer reuse, %.3f at the end\n", E->t_gpu, E->t_commit, E->t_zbatch, g_ow.t_format, g_ow.t_write, E->t_io[0], E->t_io[1]);
		fprintf(stderr, "[wtzmo-mi355x] commit sections: candidate rows + closed filter + order %.3f, window depth + seed weights %.3f, hits %.3f; planning the pairs of the ranges %.3f\n", E->t_cq[0], E->t_cq[1], E->t_cq[2], E->t_cq[3]);
		fprintf(stderr, "[wtzmo-mi355x] queries prepared ahead of the commit by helper threads: %llu taken, %llu prepared again (their read's closed pairs had moved), %.3f s waited for a helper, %.3f s starting / joining them; merging the queries' results into the global state (flush) %.3f s; the sections below count the committing thread only\n", (unsigned long long)E->n_spec_hit, (unsigned long long)E->n_spec_miss, E->t_spec_wait, E->t_spec_ctl, E->t_flush);
		fprintf(stderr, "[wtzmo-mi355x] window depth + seed weights in parts: window marks %.3f, running depth %.3f, seed weights %.3f, seed order %.3f\n", E->t_cq1[0], E->t_cq1[1], E->t_cq1[2], E->t_cq1[3]);
		fprintf(stderr, "[wtzmo-mi355x] wall seconds per call: candidates %.3f pairs_seed %.3f pairs_windows %.3f pairs_align %.3f cigar_text %.3f\n", E->t_call[0], E->t_call[1], E->t_call[2], E->t_call[3], E->t_call[4]);
	if(E->n_split) fprintf(stderr, "[wtzmo-mi355x] %llu range(s) had to be split after a scratch-pool overflow (planned at %.0f KB per pair)\n", (unsigned long long)E->n_split, E->bytes_per_pair / 1024.0);
	fprintf(stderr, "[wtzmo-mi355x] %llu batches in %llu ranges on %u worker context(s); speculation: queries %llu/%llu pairs %llu/%llu alignments %llu/%llu (used/planned)\n",
			(unsigned long long)E->n_batches, (unsigned long long)E->n_ranges, E->rows_all ? 1u : E->n_workers, (unsigned long long)E->used_queries, (unsigned long long)E->spec_queries, (unsigned long long)E->used_pairs, (unsigned long long)E->spec_pairs, (unsigned long long)E->used_items, (unsigned long long)E->spec_items);
		fprintf(stderr, "[wtzmo-mi355x] kernel ms: index %.1f zindex %.1f candidates %.1f pairs %.1f winalign %.1f stitch %.1f (K-sw3 wave %.1f, K-sw2 gaps %.1f); cells shift %llu fixed %llu global %llu; pool peak %.2f GB\n",
			cn.ms_index, cn.ms_zindex, cn.ms_candidates, cn.ms_pairs, cn.ms_winalign, cn.ms_stitch, cn.ms_ext, cn.ms_gap, (unsigned long long)cn.cells_shift, (unsigned long long)cn.cells_fixed, (unsigned long long)cn.cells_global, cn.pool_peak / 1073741824.0);
		if(statsf){ FILE *sf = fopen(statsf, "a"); if(sf){ fprintf(sf, "%llu\t%llu\t%.6f\t%.6f\t%.3f\t%.3f\t%.3f\t%.3f\t%.3f\t%.3f\t%llu\t%llu\t%llu\t%llu\t%llu\t%.3f\t%llu\t%llu\t%llu\t%.3f\t%llu\t%llu\t%llu\t%.4f\t%llu\t%.4f\t%.4f\t%.4f\t%.4f\t%.4f\t%.4f\t%llu\t%llu\n", (unsigned long long)E->n_pairs, (unsigned long long)E->pair_bp, t1 - t0, t_index,
				cn.ms_index, cn.ms_zindex, cn.ms_candidates, cn.ms_pairs, cn.ms_winalign, cn.ms_stitch, (unsigned long long)cn.cells_shift, (unsigned long long)cn.cells_fixed, (unsigned long long)cn.cells_global, (unsigned long long)cn.bytes_seed_algo, (unsigned long long)E->nrec,
				cn.ms_ext, (unsigned long long)cn.n_extjobs, (unsigned long long)E->used_queries, (unsigned long long)cn.pool_peak, cn.ms_gap, (unsigned long long)E->n_ranges, (unsigned long long)E->n_split, (unsigned long long)cn.bytes_zmer_algo, E->ing_ms, (unsigned long long)E->ing_bytes,
				E->t_gpu, E->t_commit, E->t_cq[0], E->t_cq[1], E->t_cq[2], E->t_cq[3], (unsigned long long)E->n_batches, (unsigned long long)E->spec_queries); fclose(sf); } }      /* columns 25-32 (round 6): host seconds in the device-stage calls / in the commit / its four sections, batches, planned queries */
	}
	stale_join(&stale_job); if(stale_job.pending) unlink(stale_job.path);
	free(stale_job.path);
	if(write_contained && strcmp(output, "-")){
		char *maskf = (char*)hx_realloc(NULL, strlen(output) + 16); sprintf(maskf, "%s.contained", output);
		FILE *mf = fopen(maskf, "w");
		for(uint32_t i = 0; i < n_rd; i++) if(E->masked[i]) fprintf(mf, "%s\n", E->st.reads[i].name);
		fclose(mf); free(maskf);
	}
	if(pairoutf){
		/* the reference lists the pairs in the iteration order of its hash set: replay the insertions (wtz_host.h, hx_refslots) */
		FILE *pf = fopen(pairoutf, "w");
		hx_refslots_t T; hx_refslots_init(&T, 1023);                 /* init_u64hash(1023), wtzmo.c:138 */
		for(size_t i = 0; i < E->n_order; i++){ if(E->closed_order[i] & 1u) hx_refslots_touch(&T); else hx_refslots_put(&T, E->closed_order[i]); }
		for(uint64_t k = 0; k < T.size; k++) if(T.full[k]) fprintf(pf, "%s\t%s\n", E->st.reads[(uint32_t)(T.slot[k] >> 33)].name, E->st.reads[(uint32_t)((T.slot[k] & 0xFFFFFFFFu) >> 1)].name);
		fclose(pf); free(T.slot); free(T.full);
	}
	for(int w = 0; w < 16; w++) wtz_host_free(E->cig_keep[w]);
	for(uint32_t d = 0; d < E->ndev; d++) wtz_ctx_destroy(E->ctxs[d]);
	free(E->cq.cand); free(E->cq.seeds); free(E->cq.dep);
	if(E->spec){ for(uint32_t k = 0; k < CQ_RING; k++){ free(E->spec->ring[k].w.cand); free(E->spec->ring[k].w.seeds); free(E->spec->ring[k].w.dep); } free(E->spec); }
	free(E->closed_touch);
	return 0;
}
