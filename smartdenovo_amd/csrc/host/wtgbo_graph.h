/*
 * wtgbo_graph.h — host-side (plain C) candidate side of the drop-in `wtgbo` executable: the overlap graph of
 * wtlay.h as far as wtgbo.c uses it, and the two candidate walks of wtgbo.c.  Everything here is sequential bookkeeping over
 * integers; the pair work the candidates feed (align_hzmaux, hzm_aln.h:1684-1775) runs on the device through include/wtzmo_hip.h.
 *
 *   gb_fields_from_text / gb_fields_from_record (wtgbo_core.h) + gb_accept_overlap  <- parse_overlap_item_strgraph  wtlay.h:238-274
 *   gb_biedge_of            <- overlap_item2biedge(_v2)_strgraph      wtlay.h:276-344
 *   gb_load_overlaps        <- load_overlaps_strgraph                 wtlay.h:443-468
 *   gb_build_edges          <- load_overlaps_core_strgraph            wtlay.h:349-441
 *   gb_edge_coverage        <- cal_edge_coverage_strgraph             wtlay.h:640-680
 *   gb_drop_duplicate_edges <- remove_duplicate_edges_strgraph        wtlay.h:601-638
 *   gb_mask_contained       <- mask_contained_reads_strgraph          wtlay.h:697-766
 *   gb_mask_low_cov         <- mask_low_cov_edge_strgraph             wtlay.h:682-695
 *   gb_best_overlap         <- best_overlap_strgraph                  wtlay.h:768-829
 *   gb_graph_candidates     <- collect_graph_candidates_wtgbo         wtgbo.c:62-118
 *   gb_anchor_candidates    <- collect_anchor_candidates_wtgbo        wtgbo.c:216-265
 *
 * What is observable and therefore restated exactly: the order of the edges of a node (= order of the accepted overlaps), the
 * bit-field widths of an edge (off:20, cov:6, ol_var int16, rev_idx uint16), the heap of the graph walk (list.h:78-144, ties on the
 * offset), the unstable sort of the anchor marks (sort.h:104-155) and the slot order of the `uuhash` the anchor walk iterates
 * (hashset.h: linear probing, deletions as tombstones, growth by in-place re-insertion).  Sets that are only asked "is it there"
 * (wtgbo.c:65,93-95; wtlay.h:607,646) are stamp arrays here.
 */
#ifndef WTGBO_GRAPH_H
#define WTGBO_GRAPH_H

#include "wtz_host.h"

#define GB_MAX_EDGE 0x3FFu          /* SG_MAX_EDGE, wtlay.h:35 */
#define GB_MAX_COV  63u             /* SG_EDGE_MAX_COV, wtlay.h:37 */
#define GB_TRACE_LEVEL 2u           /* WTGBO_GRAPH_TRACE_LEVEL, wtgbo.c:60 */

typedef struct { uint32_t node[2]; int32_t off[2], ol[2]; int8_t dir[2]; int32_t score; } gb_biedge_t;      /* sg_biedge_t, wtlay.h:48-54 */
typedef struct { gb_biedge_t *a; size_t n, cap; } gb_biedges_t;
static inline void gb_biedges_push(gb_biedges_t *v, const gb_biedge_t *b){
	if(v->n == v->cap){ v->cap = v->cap ? v->cap * 2 : 1024; v->a = (gb_biedge_t*)hx_realloc(v->a, sizeof(gb_biedge_t) * v->cap); }
	v->a[v->n++] = *b;
}

typedef struct {                    /* sg_edge_t, wtlay.h:57-63; the widths of the reference's bit-fields are kept by masking on store */
	uint32_t node; int32_t score;
	uint32_t off;                   /* :20 */
	int16_t  ol_var; uint16_t rev_idx;
	uint8_t  dir, closed, att, tta, cov;
} gb_edge_t;

typedef struct { uint64_t eoff[2]; uint32_t ecnt[2]; uint8_t mutual[2]; } gb_node_t;   /* mutual[k] = bogs[1][k][0] (wtlay.h:820-823): living out-edges whose partner lives */

typedef struct { uint32_t node[2]; int dir[2], beg[2], end[2], score, identity; } gb_ovl_t;               /* OverlapData, wtlay.h:229-236 */

typedef struct {
	uint32_t n_rd;
	const uint32_t *rdlen;          /* clipped lengths, read-id (= file) order */
	int min_score; float min_id; int max_margin; int mat_score;
	gb_node_t *nodes;
	uint8_t *dead;                  /* node_status */
	gb_edge_t *edges; uint64_t n_edges, cap_edges;
	/* stamp sets */
	uint32_t *stamp_a, *val_a; uint32_t cur_a;     /* per node id */
	uint32_t *stamp_b; uint32_t cur_b;             /* per (node << 1 | dir) */
} gb_graph_t;

static void gb_graph_init(gb_graph_t *g, uint32_t n_rd, const uint32_t *rdlen){
	memset(g, 0, sizeof *g);
	g->n_rd = n_rd; g->rdlen = rdlen;
	g->min_score = 200; g->min_id = 0.6f; g->max_margin = 500; g->mat_score = 0;      /* init_strgraph, wtlay.h:110-113 */
	g->nodes = (gb_node_t*)calloc((size_t)n_rd + 1, sizeof(gb_node_t));
	g->dead = (uint8_t*)calloc((size_t)n_rd + 1, 1);
	g->stamp_a = (uint32_t*)calloc((size_t)n_rd + 1, 4); g->val_a = (uint32_t*)calloc((size_t)n_rd + 1, 4);
	g->stamp_b = (uint32_t*)calloc(((size_t)n_rd + 1) * 2, 4);
	if(!g->nodes || !g->dead || !g->stamp_a || !g->val_a || !g->stamp_b){ fprintf(stderr, " -- Out of memory --\n"); exit(1); }
}
static inline uint32_t gb_next_a(gb_graph_t *g){ if(++g->cur_a == 0){ memset(g->stamp_a, 0, ((size_t)g->n_rd + 1) * 4); g->cur_a = 1; } return g->cur_a; }
static inline uint32_t gb_next_b(gb_graph_t *g){ if(++g->cur_b == 0){ memset(g->stamp_b, 0, ((size_t)g->n_rd + 1) * 8); g->cur_b = 1; } return g->cur_b; }

/* One overlap as its consumer sees it, whatever carried it here: a text line already cut into its tab-separated, non-empty columns
 * (split_string, string.h:251-274) or a binary record (include/wtz_ovlb.h).  Both loaders fill this and gb_accept_overlap decides. */
typedef struct {
	uint32_t node[2];            /* read ids in THIS program's numbering, 0xFFFFFFFF = name unknown here */
	const char *name[2];         /* for the message only */
	int dir[2], len[2], beg[2], end[2];
	int score, mat;              /* columns 11 and 13 */
	int identity;                /* (int)(atof(column 12) * 1000): the PRINTED three decimals, not mat / aln */
} gb_ovl_fields_t;

/* The filter in front of the graph (what parse_overlap_item_strgraph, wtlay.h:238-274, lets through).  1 = usable, 0 = skip.
 * The ORDER of the tests is observable - a read-length mismatch ends the program, everything else skips the overlap - so side 0 is
 * judged completely before side 1 is looked at, and the score / identity thresholds come last. */
static int gb_accept_overlap(const gb_graph_t *g, const gb_ovl_fields_t *f, gb_ovl_t *d){
	for(int s = 0; s < 2; s++){
		const uint32_t id = f->node[s];
		if(id == 0xFFFFFFFFu) return 0;
		if(s == 1 && id == f->node[0]) return 0;               /* a read against itself */
		if(g->rdlen[id] == 0) return 0;
		if(f->len[s] != (int)g->rdlen[id]){
			fprintf(stderr, " -- overlap file and read file disagree: read %s is %d bp long here, the overlap says %d --\n", f->name[s], (int)g->rdlen[id], f->len[s]);
			fflush(NULL); _exit(1);
		}
		d->node[s] = id; d->dir[s] = f->dir[s]; d->beg[s] = f->beg[s]; d->end[s] = f->end[s];
	}
	d->score = g->mat_score ? f->mat : f->score;
	d->identity = f->identity;
	if(d->score < g->min_score) return 0;
	if((float)d->identity < 1000 * g->min_id) return 0;
	return 1;
}
/* a text line: needs its first 16 columns (wtlay.h:239-241) */
static int gb_fields_from_text(const hx_names_t *names, char **col, int ncol, gb_ovl_fields_t *f){
	if(ncol < 16) return 0;
	for(int s = 0; s < 2; s++){
		char **c = col + 5 * s;       /* name, strand, length, begin, end */
		f->name[s] = c[0]; f->node[s] = hx_names_get(names, c[0]);
		f->dir[s] = (c[1][0] == '-'); f->len[s] = atoi(c[2]); f->beg[s] = atoi(c[3]); f->end[s] = atoi(c[4]);
	}
	f->score = atoi(col[10]); f->identity = (int)(atof(col[11]) * 1000); f->mat = atoi(col[12]);
	return 1;
}

/* check_dead = the loader's form (wtlay.h:276-310); without it the form wtgbo applies to its own hits (wtlay.h:312-344) */
static int gb_biedge_of(const gb_graph_t *g, const gb_ovl_t *d, gb_biedge_t *e, int check_dead){
	if(check_dead && (g->dead[d->node[0]] || g->dead[d->node[1]])) return 0;
	const int len1 = (int)g->rdlen[d->node[0]], len2 = (int)g->rdlen[d->node[1]];
	const int l0 = d->beg[0], l1 = d->beg[1], r0 = len1 - d->end[0], r1 = len2 - d->end[1];
	const int lm = l0 < l1 ? l0 : l1, rm = r0 < r1 ? r0 : r1;
	if(lm + rm > g->max_margin) return 0;
	const int a = (l0 >= l1) ? 0 : 1, b = 1 - a;         /* the read that starts further left comes first */
	e->node[0] = d->node[a]; e->node[1] = d->node[b];
	e->dir[0] = (int8_t)d->dir[a]; e->dir[1] = (int8_t)d->dir[b];
	e->off[0] = (a == 0 ? l0 : l1) - lm;
	e->off[1] = (a == 0 ? r1 : r0) - rm;
	e->ol[0] = d->end[a] - d->beg[a]; e->ol[1] = d->end[b] - d->beg[b];
	e->score = d->score;
	return 1;
}

/* the per-node edge counters admit at most GB_MAX_EDGE edges per side (wtlay.h:458-462, wtgbo.c:545-550) */
static inline int gb_count_biedge(gb_graph_t *g, const gb_biedge_t *b){
	gb_node_t *n0 = &g->nodes[b->node[0]], *n1 = &g->nodes[b->node[1]];
	const int k0 = b->dir[0], k1 = !b->dir[1];
	if(n0->ecnt[k0] >= GB_MAX_EDGE) return 0;
	if(n1->ecnt[k1] >= GB_MAX_EDGE) return 0;
	n0->ecnt[k0]++; n1->ecnt[k1]++;
	return 1;
}

/* wtlay.h:349-441: edge lists laid out side 0 of every node, then side 1 of every node; two directed edges per overlap */
static void gb_build_edges(gb_graph_t *g, const gb_biedges_t *bs){
	uint64_t off = 0;
	for(int k = 0; k < 2; k++) for(uint32_t i = 0; i < g->n_rd; i++){ gb_node_t *n = &g->nodes[i]; n->eoff[k] = off; off += n->ecnt[k]; n->ecnt[k] = 0; }
	if(off + 1 > g->cap_edges){ g->cap_edges = off + 1 + off / 2; g->edges = (gb_edge_t*)hx_realloc(g->edges, sizeof(gb_edge_t) * g->cap_edges); }
	g->n_edges = off;
	memset(g->edges, 0, sizeof(gb_edge_t) * (off + 1));
	for(size_t i = 0; i < bs->n; i++){
		const gb_biedge_t *b = &bs->a[i];
		const int len1 = (int)g->rdlen[b->node[0]], len2 = (int)g->rdlen[b->node[1]];
		gb_node_t *n = &g->nodes[b->node[0]];
		int k = b->dir[0];
		const uint32_t i0 = n->ecnt[k];
		gb_edge_t *e1 = &g->edges[n->eoff[k] + n->ecnt[k]]; n->ecnt[k]++;
		e1->node = b->node[1]; e1->dir = (uint8_t)(b->dir[1] & 1); e1->closed = 0;
		e1->off = (uint32_t)b->off[0] & 0xFFFFFu;
		int len = ((int)e1->off + len2 > len1) ? len1 - (int)e1->off : len2;
		e1->ol_var = (int16_t)(b->ol[0] - len); e1->score = b->score; e1->att = 0; e1->tta = 0;
		n = &g->nodes[b->node[1]];
		k = !b->dir[1];
		const uint32_t i1 = n->ecnt[k];
		gb_edge_t *e2 = &g->edges[n->eoff[k] + n->ecnt[k]]; n->ecnt[k]++;
		e2->node = b->node[0]; e2->dir = (uint8_t)(!b->dir[0]); e2->closed = 0;
		e2->off = (uint32_t)b->off[1] & 0xFFFFFu;
		len = ((int)e2->off + len1 > len2) ? len2 - (int)e2->off : len1;
		e2->ol_var = (int16_t)(b->ol[1] - len); e2->score = b->score; e2->att = 0; e2->tta = 0;
		e1->rev_idx = (uint16_t)i1; e2->rev_idx = (uint16_t)i0;
		/* containment marks: `att` on the edge that leaves the contained read (wtlay.h:414-438) */
		int first_in_second;       /* 1: e1->att (read 0 lies inside read 1), 0: e2->att, -1: neither */
		if(b->off[0] == 0){
			if(b->off[1] == 0){
				if(len1 < len2) first_in_second = 1;
				else if(len1 > len2) first_in_second = 0;
				else first_in_second = (b->node[0] < b->node[1]) ? 0 : 1;
			} else first_in_second = 1;
		} else if(b->off[1] == 0) first_in_second = 0;
		else first_in_second = -1;
		if(first_in_second == 1){ e1->att = 1; e2->tta = 1; }
		else if(first_in_second == 0){ e2->att = 1; e1->tta = 1; }
	}
}

static inline gb_edge_t *gb_edge(gb_graph_t *g, uint32_t node, int k, uint32_t idx){ return &g->edges[g->nodes[node].eoff[k] + idx]; }
static inline gb_edge_t *gb_partner(gb_graph_t *g, uint32_t node, int k, uint32_t idx){ gb_edge_t *e = gb_edge(g, node, k, idx); return gb_edge(g, e->node, !e->dir, e->rev_idx); }
static inline void gb_cut_both(gb_graph_t *g, uint32_t node, int k, uint32_t idx){ gb_edge(g, node, k, idx)->closed = 1; gb_partner(g, node, k, idx)->closed = 1; }   /* cut_biedge_strgraph, wtlay.h:506-514 */

/* edge_overlap_strgraph, wtlay.h:559-567 */
static inline uint32_t gb_edge_overlap(gb_graph_t *g, uint32_t node, int k, uint32_t idx){
	const gb_edge_t *e = gb_edge(g, node, k, idx);
	const int len1 = (int)g->rdlen[node], len2 = (int)g->rdlen[e->node];
	const int len = ((int)e->off + len2 > len1) ? len1 - (int)e->off : len2;
	return (uint32_t)(len + e->ol_var);
}

/* wtlay.h:640-680: cov of an edge = number of (non-closed) edges of its target that lead to a neighbour of the source */
static void gb_edge_coverage(gb_graph_t *g){
	for(uint64_t I = 0; I < g->n_edges; I++) g->edges[I].cov = GB_MAX_COV;
	for(uint32_t i = 0; i < g->n_rd; i++){
		gb_node_t *n = &g->nodes[i];
		const uint32_t st = gb_next_a(g);
		for(int k = 0; k < 2; k++) for(uint32_t j = 0; j < n->ecnt[k]; j++){ gb_edge_t *e = gb_edge(g, i, k, j); if(e->closed == 1) continue; g->stamp_a[e->node] = st; }
		for(int k = 0; k < 2; k++) for(uint32_t j = 0; j < n->ecnt[k]; j++){
			gb_edge_t *e = gb_edge(g, i, k, j);
			if(e->closed == 1) continue;
			if(e->cov != GB_MAX_COV) continue;
			uint32_t cov = 0;
			gb_node_t *n2 = &g->nodes[e->node];
			for(int k2 = 0; k2 < 2; k2++) for(uint32_t j2 = 0; j2 < n2->ecnt[k2]; j2++){
				gb_edge_t *e2 = gb_edge(g, e->node, k2, j2);
				if(e2->closed == 1) continue;
				if(g->stamp_a[e2->node] == st) cov++;
			}
			if(cov + 1 >= GB_MAX_COV) cov = GB_MAX_COV - 1;
			e->cov = (uint8_t)cov;
			g->edges[n2->eoff[!e->dir] + e->rev_idx].cov = (uint8_t)cov;
		}
	}
}

/* wtlay.h:601-638: of two living edges of one side to the same neighbour the better-scoring one stays (the later one on a tie) */
static uint64_t gb_drop_duplicate_edges(gb_graph_t *g){
	uint64_t ret = 0;
	for(uint32_t i = 0; i < g->n_rd; i++){
		if(g->dead[i]) continue;
		gb_node_t *n = &g->nodes[i];
		for(int k = 0; k < 2; k++){
			const uint32_t st = gb_next_a(g);
			for(uint32_t j = 0; j < n->ecnt[k]; j++){
				gb_edge_t *e = gb_edge(g, i, k, j);
				if(e->closed) continue;
				if(g->stamp_a[e->node] == st){
					ret++;
					gb_edge_t *e2 = gb_edge(g, i, k, g->val_a[e->node]);
					if(e->score < e2->score) gb_cut_both(g, i, k, j);
					else { gb_cut_both(g, i, k, g->val_a[e->node]); g->val_a[e->node] = j; }
				} else { g->stamp_a[e->node] = st; g->val_a[e->node] = j; }
			}
		}
	}
	return ret;
}

static uint64_t gb_mask_low_cov(gb_graph_t *g, uint32_t cutoff){         /* wtlay.h:682-695 */
	uint64_t ret = 0;
	if(cutoff == 0) return 0;
	for(uint64_t i = 0; i < g->n_edges; i++){ gb_edge_t *e = &g->edges[i]; if(e->closed == 1) continue; if(e->cov >= cutoff) continue; e->closed = 1; ret++; }
	return ret;
}

/* wtlay.h:697-766: a read with a living `att` edge is contained; it keeps the att mark only towards its container
 * (the best-scoring uncontained one, else the first contained one), then all its edges are cut and it is dead */
static uint32_t gb_mask_contained(gb_graph_t *g){
	uint32_t ret = 0;
	uint8_t *flag = (uint8_t*)calloc((size_t)g->n_rd + 1, 1);
	for(uint32_t i = 0; i < g->n_rd; i++){
		if(g->dead[i]) continue;
		gb_node_t *n = &g->nodes[i];
		int c = 0;
		for(int k = 0; c == 0 && k < 2; k++) for(uint32_t j = 0; j < n->ecnt[k]; j++){ gb_edge_t *e = gb_edge(g, i, k, j); if(e->closed == 1) continue; if(e->att){ c = 1; break; } }
		if(!c) continue;
		flag[i] = 1; ret++;
	}
	for(uint32_t i = 0; i < g->n_rd; i++){
		if(g->dead[i] || !flag[i]) continue;
		gb_node_t *n = &g->nodes[i];
		uint32_t c = 0xFFFFFFFFu; int max_score = 0;
		for(int k = 0; k < 2; k++) for(uint32_t j = 0; j < n->ecnt[k]; j++){
			gb_edge_t *e = gb_edge(g, i, k, j);
			if(e->closed == 1 || !e->att) continue;
			if(flag[e->node]){ if(c == 0xFFFFFFFFu) c = e->node; continue; }
			if(e->score > max_score){ c = e->node; max_score = e->score; }
		}
		for(int k = 0; k < 2; k++) for(uint32_t j = 0; j < n->ecnt[k]; j++){ gb_edge_t *e = gb_edge(g, i, k, j); if(e->node != c) e->att = 0; }
	}
	for(uint32_t i = 0; i < g->n_rd; i++){
		if(!flag[i]) continue;
		gb_node_t *n = &g->nodes[i];
		for(int k = 0; k < 2; k++) for(uint32_t j = 0; j < n->ecnt[k]; j++) gb_cut_both(g, i, k, j);      /* mask_node_strgraph, wtlay.h:577-587 */
		g->dead[i] = 1;
	}
	free(flag);
	return ret;
}

/* wtlay.h:768-829: per side the edge with the smallest offset among those scoring >= cutoff x the side's best stays;
 * then mutual[k] counts the living edges whose reverse edge lives too */
static uint64_t gb_best_overlap(gb_graph_t *g, float cutoff){
	uint64_t ret = 0;
	for(uint32_t i = 0; i < g->n_rd; i++){
		if(g->dead[i]) continue;
		gb_node_t *n = &g->nodes[i];
		for(int k = 0; k < 2; k++){
			float bestS = 0;
			for(uint32_t j = 0; j < n->ecnt[k]; j++){ gb_edge_t *e = gb_edge(g, i, k, j); if(e->closed || e->att || e->tta) continue; if((float)e->score > bestS) bestS = (float)e->score; }
			bestS = bestS * cutoff;
			int best = (int)g->rdlen[i]; uint32_t b = 0xFFFFFFFFu;
			for(uint32_t j = 0; j < n->ecnt[k]; j++){
				gb_edge_t *e = gb_edge(g, i, k, j);
				if(e->closed || e->att || e->tta) continue;
				if((float)e->score < bestS) continue;
				if((int)e->off < best){ best = (int)e->off; b = j; }
			}
			for(uint32_t j = 0; j < n->ecnt[k]; j++) if(j != b){ gb_edge(g, i, k, j)->closed = 1; ret++; }
		}
	}
	for(uint32_t i = 0; i < g->n_rd; i++){ g->nodes[i].mutual[0] = g->nodes[i].mutual[1] = 0; }
	for(uint32_t i = 0; i < g->n_rd; i++){
		if(g->dead[i]) continue;
		gb_node_t *n = &g->nodes[i];
		for(int k = 0; k < 2; k++) for(uint32_t j = 0; j < n->ecnt[k]; j++){
			gb_edge_t *e = gb_edge(g, i, k, j);
			if(e->closed) continue;
			if(!gb_partner(g, i, k, j)->closed){ if(n->mutual[k] < 0xFFu) n->mutual[k]++; }
		}
	}
	return ret;
}

/* ---------------- the closed-pair set: membership + the reference's slot order for the -9 file ---------------- */
typedef struct { hx_set_t has; hx_refslots_t slots; } gb_closed_t;
static void gb_closed_init(gb_closed_t *c){ memset(c, 0, sizeof *c); hx_refslots_init(&c->slots, 1023); }        /* init_u64hash(1023), wtgbo.c:487 */
/* put_u64hash / prepare_u64hash: both run the capacity check first (hashset.h:131,351).  1 = the key is new */
static int gb_closed_put(gb_closed_t *c, uint64_t key){
	if(hx_set_put(&c->has, key)){ hx_refslots_put(&c->slots, key); return 1; }
	hx_refslots_touch(&c->slots); return 0;
}
static inline uint64_t gb_pair_id(uint64_t a, uint64_t b, uint64_t dir){ return a < b ? ((a << 33) | (b << 1) | dir) : ((b << 33) | (a << 1) | dir); }    /* ovl_uniq_long_id, wtlay.h:346-347 */

/* ---------------- min-heap of the graph walk: array_heap_push / array_heap_pop, list.h:78-144, key = value >> 40 ---------------- */
typedef struct { uint64_t *a; size_t n, cap; } gb_heap_t;
static void gb_heap_push(gb_heap_t *h, uint64_t v){
	if(h->n == h->cap){ h->cap = h->cap ? h->cap * 2 : 1024; h->a = (uint64_t*)hx_realloc(h->a, 8 * h->cap); }
	size_t i = h->n; h->a[h->n++] = v;
	while(i){
		const size_t j = (i - 1) >> 1;
		if((h->a[i] >> 40) >= (h->a[j] >> 40)) break;
		const uint64_t t = h->a[i]; h->a[i] = h->a[j]; h->a[j] = t;
		i = j;
	}
}
static uint64_t gb_heap_pop(gb_heap_t *h){
	const uint64_t top = h->a[0];
	size_t idx = 0;
	h->a[0] = h->a[--h->n];
	while((idx << 1) + 1 < h->n){
		size_t sw = idx;
		if((h->a[sw] >> 40) > (h->a[(idx << 1) + 1] >> 40)) sw = (idx << 1) + 1;
		if((idx << 1) + 2 < h->n && (h->a[sw] >> 40) > (h->a[(idx << 1) + 2] >> 40)) sw = (idx << 1) + 2;
		if(sw == idx) break;
		const uint64_t t = h->a[idx]; h->a[idx] = h->a[sw]; h->a[sw] = t;
		idx = sw;
	}
	return top;
}

typedef struct { uint32_t *a; size_t n, cap; } gb_u32v_t;
static inline void gb_u32v_push(gb_u32v_t *v, uint32_t x){ if(v->n == v->cap){ v->cap = v->cap ? v->cap * 2 : 1024; v->a = (uint32_t*)hx_realloc(v->a, 4 * v->cap); } v->a[v->n++] = x; }
typedef struct { uint64_t *a; size_t n, cap; } gb_u64v_t;
static inline void gb_u64v_push(gb_u64v_t *v, uint64_t x){ if(v->n == v->cap){ v->cap = v->cap ? v->cap * 2 : 1024; v->a = (uint64_t*)hx_realloc(v->a, 8 * v->cap); } v->a[v->n++] = x; }

/* wtgbo.c:62-118: reads reachable from `node_id` within two graph steps on either side and within its own length; the ones never
 * paired with it before become candidates `read << 1 | strand` (and are entered into the closed set on the spot).
 * Edges are walked whatever their `closed` state.  After the last edge of a side whose list held an `att` edge the reference reads
 * ONE MORE element of the global edge array (wtgbo.c:81-93: `i == edge_cnts[k]` with att != 0 takes the else branch) — the first edge
 * of the next non-empty list; the same element is read here.  One past the end of the whole array the reference reads memory of its
 * edge vector it never wrote (zero-filled as it comes from calloc / fresh pages): a zero edge stands in for it and *oob is raised. */
static void gb_graph_candidates(gb_graph_t *g, uint32_t node_id, uint32_t max_ext, gb_closed_t *closed, gb_u32v_t *cands, gb_heap_t *heap, int *oob){
	const uint32_t max = max_ext + g->rdlen[node_id];
	const uint32_t st = gb_next_b(g);
	g->stamp_b[((size_t)node_id << 1) | 0] = st; g->stamp_b[((size_t)node_id << 1) | 1] = st;      /* wtgbo.c:177-178 */
	heap->n = 0;
	gb_heap_push(heap, (uint64_t)((node_id << 2) | (0u << 1) | 0u));
	gb_heap_push(heap, (uint64_t)((node_id << 2) | (1u << 1) | 1u));
	int f = 0;
	while(heap->n){
		const uint64_t idx = gb_heap_pop(heap);
		const uint32_t nid = (uint32_t)(idx & 0xFFFFFFFFu) >> 2, dir = (uint32_t)(idx >> 1) & 1u, k = (uint32_t)idx & 1u;
		const uint32_t lv = (uint32_t)(idx >> 32) & 0xFFu, off1 = (uint32_t)(idx >> 40);
		const gb_node_t *n1 = &g->nodes[nid];
		int att = 0;
		for(uint32_t i = 0; i <= n1->ecnt[k]; i++){
			const gb_edge_t *e;
			if(i == n1->ecnt[k] && att == 0){
				/* a contained read is left through the edge to its container on the other side */
				if(!g->dead[nid]) break;
				uint32_t j;
				e = NULL;
				for(j = 0; j < n1->ecnt[!k]; j++){ e = gb_edge(g, nid, !k, j); if(e->att) break; }
				if(j == n1->ecnt[!k]) break;
			} else {
				const uint64_t at = n1->eoff[k] + i;
				if(at >= g->n_edges && oob) *oob = 1;        /* gb_build_edges keeps one zero element behind the last edge */
				e = &g->edges[at];
			}
			if(e->att) att = 1;
			const uint32_t off2 = off1 + e->off;
			if(off2 > max) continue;
			const uint32_t val = (e->node << 1) | (dir ^ e->dir);
			if(g->stamp_b[val] == st) continue;
			g->stamp_b[val] = st;
			if(f && !g->dead[e->node]){
				if(gb_closed_put(closed, gb_pair_id(node_id, e->node, (uint64_t)(e->dir ^ dir)))) gb_u32v_push(cands, val);
			}
			if(lv < GB_TRACE_LEVEL) gb_heap_push(heap, ((uint64_t)off2 << 40) | (((uint64_t)lv + 1) << 32) | (uint64_t)((e->node << 2) | (dir << 1) | e->dir));
		}
		f = 1;
	}
}

/* ---------------- the (read -> strand) table of the anchor walk with the reference's slot order ---------------- */
typedef struct { uint32_t *key, *val; uint8_t *flag; uint64_t size, ocp, count, max; } gb_uu_t;      /* flag: 0 empty, 1 live, 2 deleted */
static inline uint32_t gb_hash32(uint32_t key){       /* __lh3_Jenkins_hash_int, hashset.h:438-448 */
	key += (key << 12); key ^= (key >> 22); key += (key << 4); key ^= (key >> 9); key += (key << 10); key ^= (key >> 2); key += (key << 7); key ^= (key >> 12);
	return key;
}
static void gb_uu_init(gb_uu_t *t, uint64_t hint){
	t->size = hx_ref_size_for(hint); t->ocp = t->count = 0; t->max = (uint64_t)((float)t->size * 0.67f);
	t->key = (uint32_t*)calloc(t->size, 4); t->val = (uint32_t*)calloc(t->size, 4); t->flag = (uint8_t*)calloc(t->size, 1);
}
static void gb_uu_clear(gb_uu_t *t){ if(t->ocp == 0) return; memset(t->flag, 0, t->size); t->count = 0; t->ocp = 0; }      /* hashset.h:329-336 */
static void gb_uu_encap(gb_uu_t *t){                  /* encap(set, 1), hashset.h:383-428 */
	if(t->ocp + 1 <= t->max) return;
	uint64_t n = t->size;
	do { n = hx_ref_size_for(n * 2); } while((float)n * 0.67f < (float)(t->count + 1));
	const uint64_t old = t->size;
	t->key = (uint32_t*)hx_realloc(t->key, 4 * n); t->val = (uint32_t*)hx_realloc(t->val, 4 * n);
	uint8_t *waiting = t->flag;                       /* live entries still at their old slot */
	uint8_t *flag = (uint8_t*)calloc(n, 1);
	t->size = n; t->ocp = t->count; t->max = (uint64_t)((float)n * 0.67f);
	for(uint64_t i = 0; i < old; i++){
		if(waiting[i] != 1) continue;
		uint32_t key = t->key[i], val = t->val[i]; waiting[i] = 2;
		for(;;){
			uint64_t h = gb_hash32(key) % n;
			while(flag[h]) h = (h + 1) % n;
			flag[h] = 1;
			if(h < old && waiting[h] == 1){ const uint32_t k2 = t->key[h], v2 = t->val[h]; t->key[h] = key; t->val[h] = val; key = k2; val = v2; waiting[h] = 2; }
			else { t->key[h] = key; t->val[h] = val; break; }
		}
	}
	free(waiting);
	t->flag = flag;
}
static void gb_uu_put(gb_uu_t *t, uint32_t key, uint32_t val){        /* put = encap + add, hashset.h:186-227 */
	gb_uu_encap(t);
	uint64_t h = gb_hash32(key) % t->size, d = t->size;
	for(;;){
		if(t->flag[h] == 0){
			if(d == t->size){ d = h; t->ocp++; }
			t->flag[d] = 1; t->key[d] = key; t->val[d] = val; t->count++;
			return;
		} else if(t->flag[h] == 2){ if(d == t->size) d = h; }
		else if(t->key[h] == key){ t->val[h] = val; return; }
		h = (h + 1 == t->size) ? 0 : h + 1;
	}
}
static int gb_uu_remove(gb_uu_t *t, uint32_t key){                   /* hashset.h:233-255 */
	uint64_t h = gb_hash32(key) % t->size;
	for(;;){
		if(t->flag[h] == 0) return 0;
		if(t->flag[h] == 1 && t->key[h] == key){ t->count--; t->flag[h] = 2; return 1; }
		h = (h + 1) % t->size;
	}
}

typedef struct { uint32_t node, dir, end; int pos; } gb_mark_t;         /* mark_t, wtgbo.c:210-214 */
static int gb_gt_mark(const void *a, const void *b, void *ctx){ (void)ctx; return ((const gb_mark_t*)a)->pos > ((const gb_mark_t*)b)->pos; }   /* wtgbo.c:244 */
typedef struct { gb_mark_t *a; size_t n, cap; } gb_marks_t;
static inline void gb_marks_push(gb_marks_t *v, uint32_t node, uint32_t dir, uint32_t end, int pos){
	if(v->n == v->cap){ v->cap = v->cap ? v->cap * 2 : 1024; v->a = (gb_mark_t*)hx_realloc(v->a, sizeof(gb_mark_t) * v->cap); }
	gb_mark_t m; m.node = node & 0x3FFFFFFFu; m.dir = dir & 1u; m.end = end; m.pos = pos; v->a[v->n++] = m;
}

/* wtgbo.c:216-265: the overlaps of `node_id` with living reads as intervals on it; two reads whose intervals intersect and that
 * were never paired become a candidate `pair id` (smaller read first, strand = xor of the two edge strands) */
static void gb_anchor_candidates(gb_graph_t *g, uint32_t node_id, gb_closed_t *closed, gb_u64v_t *cands, gb_marks_t *marks, gb_uu_t *tab){
	const gb_node_t *n = &g->nodes[node_id];
	const int len = (int)g->rdlen[node_id];
	marks->n = 0;
	for(int k = 0; k < 2; k++) for(uint32_t i = 0; i < n->ecnt[k]; i++){
		const gb_edge_t *e = gb_edge(g, node_id, k, i);
		if(g->dead[e->node]) continue;
		int beg, end;
		const int ovl = (int)gb_edge_overlap(g, node_id, k, i);
		if(k){ beg = len - ((int)e->off + ovl); end = len - (int)e->off; }
		else { beg = (int)e->off; end = (int)e->off + ovl; }
		const uint32_t dir = e->dir ^ (uint32_t)k;
		gb_marks_push(marks, e->node, dir, 0, beg);
		gb_marks_push(marks, e->node, dir, 1, end);
	}
	hx_sort_exact(marks->a, marks->n, sizeof(gb_mark_t), gb_gt_mark, NULL);
	gb_uu_clear(tab);
	for(size_t i = 0; i < marks->n; i++){
		const gb_mark_t *m = &marks->a[i];
		if(m->end){
			gb_uu_remove(tab, m->node);
			for(uint64_t s = 0; s < tab->size; s++){
				if(tab->flag[s] != 1) continue;
				const uint64_t key = gb_pair_id(m->node, tab->key[s], (uint64_t)(m->dir ^ tab->val[s]));
				if(gb_closed_put(closed, key)) gb_u64v_push(cands, key);
			}
		} else gb_uu_put(tab, m->node, m->dir);
	}
}

#endif
