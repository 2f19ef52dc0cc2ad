/* oracle/mc_oracle.h -- TEST INFRASTRUCTURE ONLY (see mc_oracle.c). */
#ifndef MC_ORACLE_H_
#define MC_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCO_NUM_RANKS 21 /* taxonomy.hpp:103 */

typedef struct { uint32_t win; uint32_t tgt; } mco_hit;
typedef struct { int64_t taxid; uint32_t tgt; uint32_t hits; uint32_t beg; uint32_t end; } mco_cand;

/* rows 1-5: all window sketches of one sequence */
int64_t mco_sketch(const char* seq, uint64_t len, uint32_t k, uint32_t s, uint32_t w, uint32_t stride,
                   uint32_t* feats, uint32_t* counts, uint64_t maxWindows);

/* one window [p, p + n): up to s features ascending into out, returns their count, -1 = shorter than k (no window).  _plain is
   the restatement every other entry uses, _fast the rolling form the database build uses; tests hold them against each other */
int mco_sketch_window_plain(const char* p, uint64_t n, unsigned k, uint32_t s, uint32_t* out);
int mco_sketch_window_fast(const char* p, uint64_t n, unsigned k, uint32_t s, uint32_t* out);

/* row 11-13: database files */
void* mco_db_open(const char* name);
void* mco_db_open_part(const char* name, int part);
void  mco_db_close(void* db);
void  mco_db_info(void* db, uint64_t* info8);
int   mco_db_target_id_bytes(void* db);
void  mco_db_max_locations_per_feature(void* db, uint64_t n);
uint64_t mco_db_remove_features_with_more_locations_than(void* db, uint64_t n);
void  mco_db_lineages(void* db, int64_t* out);
int64_t mco_db_target_name(void* db, uint64_t tgt, char* buf, uint64_t cap);
/* row 6: one lookup; returns bucket size, *vals -> values as (tgt<<32 | win) */
uint32_t mco_db_lookup(void* db, uint32_t part, uint32_t feature, const uint64_t** vals);

uint64_t mco_db_part_arrays(void* db, uint32_t part, const uint32_t** keys, const uint32_t** sizes, const uint64_t** offs, const uint64_t** values);

/* rows 7-10 */
void* mco_handler_new(void);
void  mco_handler_free(void* h);
/* mode 0: reference behaviour incl. the multi-part sorter ping-pong (query_handler.hpp:75-101)
   mode 1: intended multi-part semantics (per-part sorted lists concatenated) */
int mco_query(void* db, void* handler,
              const char* s1, uint64_t l1, const char* s2, uint64_t l2,
              uint32_t sketchlen, uint32_t winlen, uint32_t winstride,
              uint64_t maxCand, int lowestRank, uint64_t insertSizeMax, int mode,
              const mco_hit** allhits, uint64_t* nAll,
              const mco_cand** tophits, uint64_t* nTop);

/* rows 9-10 on an explicit location list; locs = (tgt<<32 | win) in list order.
   taxkey: per-target taxon id used for merging (NULL => sequence level, taxid = -(tgt)-1);
   a taxkey of 0 means "no taxon" (candidate skipped).  Returns the number of candidates. */
uint64_t mco_candidates(const uint64_t* locs, uint64_t n, uint32_t maxWindowsInRange, uint64_t maxCand,
                        const int64_t* taxkey, int mergeAboveSequence, mco_cand* out, uint64_t cap);

/* many single-end reads on 'threads' host threads (one handler each); returns elapsed seconds.  bench.py's cpu_baseline leg
   (kind "port") and the bulk parity checks */
double mco_query_many(void* db, const char* seqs, const uint64_t* offs, uint64_t n,
                      uint64_t maxCand, int lowestRank, uint64_t insertSizeMax, int threads,
                      mco_cand* cands);

/* the database BUILD restated (database.cpp:34-81, host_hashmap.hpp:570-605), restricted to the features in wanted[nWanted]
   (NULL = all): target t = lengths[t] characters written by gen(user, t, dst); lineage = [numTargets * 21] taxon ids or NULL;
   targetWindowsOut (may be NULL) receives the window count of every target.  Returns a database handle for mco_query & co. */
void* mco_db_build(uint32_t k, uint32_t s, uint32_t w, uint32_t stride, uint32_t maxLocs, int targetBytes,
                   uint32_t numTargets, const uint32_t* lengths, void (*gen)(void*, uint32_t, char*), void* user,
                   const uint32_t* wanted, uint64_t nWanted, const int64_t* lineage, int threads, uint64_t* targetWindowsOut);
/* the same; every thread takes `claim` consecutive targets at a time (lets a generator reuse work between neighbouring targets) */
void* mco_db_build_claim(uint32_t k, uint32_t s, uint32_t w, uint32_t stride, uint32_t maxLocs, int targetBytes,
                         uint32_t numTargets, const uint32_t* lengths, void (*gen)(void*, uint32_t, char*), void* user,
                         const uint32_t* wanted, uint64_t nWanted, const int64_t* lineage, int threads, uint32_t claim, uint64_t* targetWindowsOut);

#ifdef __cplusplus
}
#endif
#endif
