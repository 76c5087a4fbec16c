/*
 * wtz_ovlb_loaders.h - the binary overlap stream (include/wtz_ovlb.h) inside the reference's OWN two loaders:
 *     wtlay.h   parse_overlap_item_strgraph / load_overlaps_strgraph   (wtlay.h:238-268, 443-470; also what the reference's wtgbo loads with)
 *     wtclp.c   load_alignments_wtclp                                  (wtclp.c:111-180)
 * This file is glue written against the reference's data structures; it is compiled INTO a patched copy of those two sources by
 * integration/f3_patch_loaders.py (three inserted lines per file + the fix of the out-of-bounds read at wtclp.c:171-172) and tested there:
 * `wtlay -j x.ovlb` / `wtclp -i x.ovlb` must write what they write for the text form of the same records, byte for byte.
 * A file of a `-j` / `-i` list is recognised by its first eight bytes; text files of the same list are read as before.
 *
 * Both loaders fill their record from the integers of the stream - no line splitting, no atoi - and apply their own tests in their own order
 * (the order is observable: a changed read length ends the program, everything else skips the record).  The one float of the text form, the
 * identity column, is reproduced as the PRINTED three decimals (wtz_ovlb_identity_text) and parsed like column 12 would be.
 */
#ifndef WTZ_OVLB_LOADERS_H
#define WTZ_OVLB_LOADERS_H
#include "wtz_ovlb.h"

/* Is the file the reader is about to start a binary stream?  Looks at a file once, before its first byte went into the reader's buffer; the
 * eight bytes read for the test are left in that buffer, so a text file goes on from there untouched. */
static uint32_t wtz_sniffed_upto = 0;      /* files [0, this) of the current reader have been looked at */
static int wtz_fr_begins_binary(FileReader *fr, wtz_ovlb_reader_t *rd){
	if(fr->fidx >= fr->files->size || fr->fidx < wtz_sniffed_upto || fr->ptr < fr->size) return 0;
	fr_file_t *fc = ref_fr_filev(fr->files, fr->fidx);
	if(fc->file == NULL) return 0;
	wtz_sniffed_upto = fr->fidx + 1;
	fr->ptr = 0; fr->last_brk = 0;
	fr->size = (int)fread(fr->buffer, 1, 8, fc->file);
	if(fr->size != 8 || memcmp(fr->buffer, WTZ_OVLB_MAGIC, 8) != 0) return 0;
	char first8[8]; memcpy(first8, fr->buffer, 8); fr->size = 0;
	if(wtz_ovlb_open(rd, fc->file, first8) != 0){ fprintf(stderr, " -- broken binary overlap stream (header) in %s -- %s:%d --\n", __FUNCTION__, __FILE__, __LINE__); exit(1); }
	return 1;
}
/* the stream is used up: move the reader on to its next file the way its own end-of-file branch does (file_reader.c fread_line2) */
static void wtz_fr_next_file(FileReader *fr, wtz_ovlb_reader_t *rd){
	fr_file_t *fc = ref_fr_filev(fr->files, fr->fidx);
	if(fc->is_proc) pclose(fc->file); else if(fc->file != stdin) fclose(fc->file);
	fc->file = NULL;
	wtz_ovlb_close(rd);
	fr->fidx ++;
	if(fr->fidx < fr->files->size){
		fc = ref_fr_filev(fr->files, fr->fidx);
		fc->file = fc->is_proc ? popen(fc->filename, "r") : (fc->filename ? fopen(fc->filename, "r") : stdin);
	}
}

#ifdef WTZ_OVLB_FOR_WTLAY
/* 1 = *dat holds the next overlap of a binary file; 0 = the reader stands on text (or at its end): the caller's text loop takes over */
static int wtz_lay_binary_item(StringGraph *g, FileReader *fr, OverlapData *dat){
	static wtz_ovlb_reader_t rd; static int on = 0; static uint32_t *node = NULL;
	wtz_ovlb_rec_t r; char idt[32]; int rc;
	for(;;){
		if(!on){
			if(!wtz_fr_begins_binary(fr, &rd)) return 0;
			node = (uint32_t*)realloc(node, sizeof(uint32_t) * (rd.n_reads + 1));      /* the writer's read ids -> this graph's nodes, once per read */
			for(uint64_t i = 0; i < rd.n_reads; i++) node[i] = kv_get_cuhash(g->rdname2id, rd.names[i]);
			on = 1;
		}
		while((rc = wtz_ovlb_next(&rd, &r)) == 1){
			const uint32_t a = node[r.id1], b = node[r.id2];
			if(a == 0xFFFFFFFFU) continue;
			if(g->rdlens->buffer[a] == 0) continue;
			if((int)rd.rdlen[r.id1] != (int)g->rdlens->buffer[a]){
				fprintf(stderr, " -- Inconsistent read (%s) length %d != %d in %s -- %s:%d --\n", rd.names[r.id1], g->rdlens->buffer[a], (int)rd.rdlen[r.id1], __FUNCTION__, __FILE__, __LINE__);
				exit(1);
			}
			if(b == 0xFFFFFFFFU) continue;
			if(a == b) continue;
			if(g->rdlens->buffer[b] == 0) continue;
			if((int)rd.rdlen[r.id2] != (int)g->rdlens->buffer[b]){
				fprintf(stderr, " -- Inconsistent read (%s) length %d != %d in %s -- %s:%d --\n", rd.names[r.id2], g->rdlens->buffer[b], (int)rd.rdlen[r.id2], __FUNCTION__, __FILE__, __LINE__);
				exit(1);
			}
			dat->node_id[0] = a; dat->node_id[1] = b;
			dat->dir[0] = 0; dat->dir[1] = r.dir2 & 1;
			dat->beg[0] = r.tb; dat->end[0] = r.te; dat->beg[1] = r.qb; dat->end[1] = r.qe;
			dat->score = g->mat_score ? r.mat : r.score;
			wtz_ovlb_identity_text(&r, idt);
			dat->identity = atof(idt) * 1000;
			if(dat->score < g->min_score) continue;
			if(dat->identity < 1000 * g->min_id) continue;
			return 1;
		}
		if(rc < 0){ fprintf(stderr, " -- broken binary overlap stream (record) in %s -- %s:%d --\n", __FUNCTION__, __FILE__, __LINE__); exit(1); }
		wtz_fr_next_file(fr, &rd); on = 0;
	}
}
#endif

#ifdef WTZ_OVLB_FOR_WTCLP
/* the binary files at the reader's position into wt->hits / wt->seqs, record by record like the text loop of load_alignments_wtclp */
static void wtz_clp_load_binary(WTCLP *wt, float min_sm, FileReader *fr){
	wtz_ovlb_reader_t rd; wtz_ovlb_rec_t r; char idt[32]; int rc;
	while(wtz_fr_begins_binary(fr, &rd)){
		uint32_t *sid = (uint32_t*)malloc(sizeof(uint32_t) * (rd.n_reads + 1));       /* the writer's read ids -> wt->seqs, entered at a read's first use (the text loop's order) */
		for(uint64_t i = 0; i < rd.n_reads; i++) sid[i] = 0xFFFFFFFFU;
		while((rc = wtz_ovlb_next(&rd, &r)) == 1){
			wtz_ovlb_identity_text(&r, idt);
			const float sm = atof(idt);
			if(sm < min_sm) continue;
			pb_aln_t *hit = next_ref_pbalnv(wt->hits);
			int side;
			for(side = 0; side < 2; side++){
				const uint32_t wid = side ? r.id2 : r.id1;
				pb_seq_t *pb;
				if(sid[wid] == 0xFFFFFFFFU){
					cuhash_t H, *h; int exists;
					memset(&H, 0, sizeof(cuhash_t)); H.key = rd.names[wid];
					h = prepare_cuhash(wt->tag2idx, H, &exists);
					if(exists){ sid[wid] = h->val; pb = ref_pbseqv(wt->seqs, h->val); }      /* known from a file read before this one */
					else {
						h->key = strdup(rd.names[wid]);
						sid[wid] = h->val = wt->seqs->size;
						pb = next_ref_pbseqv(wt->seqs);
						memset(pb, 0, sizeof(pb_seq_t));
						pb->tag = h->key; pb->len = (int)rd.rdlen[wid];
						pb->clp_x = 0; pb->clp_y = pb->len; pb->obts[0] = 0; pb->obts[1] = pb->len;
						pb->chg = 1; pb->fix = 0; pb->closed = 0;
					}
				} else pb = ref_pbseqv(wt->seqs, sid[wid]);
				hit->sids[side] = sid[wid];
				hit->pair[side].dir = side ? (r.dir2 & 1) : 0;
				hit->pair[side].x = side ? r.qb : r.tb;
				hit->pair[side].y = side ? r.qe : r.te;
				if(hit->pair[side].dir){
					const uint32_t x = hit->pair[side].x, y = hit->pair[side].y;
					hit->pair[side].x = pb->len - y; hit->pair[side].y = pb->len - x;
				}
				if(hit->pair[side].x + wt->min_aln_len > hit->pair[side].y){ wt->hits->size --; break; }
			}
		}
		free(sid);
		if(rc < 0){ fprintf(stderr, " -- broken binary overlap stream (record) in %s -- %s:%d --\n", __FUNCTION__, __FILE__, __LINE__); exit(1); }
		wtz_fr_next_file(fr, &rd);
	}
}
#endif
#endif
