/* metacache_amd.h -- C ABI of the MI355X-native MetaCache query hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ / torch types.
 * Every entry point names the reference interface (file:line under muellan/metacache src/) it
 * replaces.  The library behind it is hand-written HIP for gfx950; there is NO CPU fallback: every
 * call that needs the GPU fails with MC_ERR_HIP if no device is usable.
 *
 * Pipeline per batch of queries (a query = one read or one read pair), details in DESIGN.md §3:
 *   plan / scan          : windows per query -> offsets
 *   sketch_lane          : one lane per short read: rolling canonical k-mers -> min-hash sketches
 *   chunk_sketch / probe : long single reads, one lane per window
 *   probe_cands          : one lane per query: bucket lookups, location list, sort, window-range candidates, top-K
 *   mid_cands            : lists of 33..256 locations, 4 / 8 / 16 lanes per query, register sort
 *   hash_cands           : lists of 65..256 locations, one wave per query, (target, window) counts in an LDS table
 *   big_filter / big_count : lists above 128 (RefSeq-scale tables): filtered by target, then counted
 *   query_wave / sort_candidates : one wave per query for everything else (and for -allhits)
 *   (semantics = reference CPU classifier, bit for bit)
 *
 * Threading: one mc_ctx per (process, device).  mc_batch_* calls are thread-safe for DISTINCT slots (one slot per host
 * thread, like one query_handler / query_host_data per thread in the reference, database_query.hpp:204-205,
 * query_batch.cuh:369-371).  Between mc_batch_submit and mc_batch_wait a slot borrows one of a few device pipes (stream +
 * workspace), so the copies and kernels of different slots overlap on the device.  mc_query_device and
 * mc_candidates_from_hits use the context's own pipe: one caller at a time.
 */
#ifndef METACACHE_AMD_H_
#define METACACHE_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes: 0 ok, >0 recoverable, <0 fatal; nothing throws or exits across this ABI
 * (reference: C++ exceptions main.cpp:65-77, CUERR exit(1) cuda_helpers.cuh:18-26) */
#define MC_OK               0
#define MC_BATCH_FULL       1   /* mc_batch_add: query does not fit, submit first (query_batch.cuh:126-132) */
#define MC_ERR_INVALID     -1
#define MC_ERR_HIP         -2
#define MC_ERR_NOMEM       -3
#define MC_ERR_IO          -4
#define MC_ERR_UNSUPPORTED -5
#define MC_ERR_STATE       -6

#define MC_NUM_RANKS 21          /* taxonomy.hpp:103 */
#define MC_MAX_SKETCH 32         /* features per window handled on device */
#define MC_MAX_WINLEN 1024       /* characters per window handled on device */

typedef struct mc_ctx mc_ctx;

/* One top candidate = match_candidate minus the host pointer (candidate_structs.hpp:80-104).
 * Unused trailing entries of a query have hits == 0 (consumers stop at hits > 0, printing.cpp:291). */
typedef struct {
    uint32_t tgt;   /* target id                         */
    uint32_t hits;  /* number of location hits in range  */
    uint32_t beg;   /* first window of the range         */
    uint32_t end;   /* last window of the range (incl.)  */
} mc_candidate;

/* One location = database::location (database.hpp:136-166), always widened to 2 x u32 on device;
 * as a little-endian u64 it reads (tgt << 32) | win, i.e. location::operator< is integer <. */
typedef struct {
    uint32_t win;
    uint32_t tgt;
} mc_location;

typedef struct {
    int32_t  device;                 /* HIP device ordinal */
    /* query sketching (sketching_options, hash_dna.hpp:99-163); kmerlen must equal the DB's */
    uint32_t kmerlen;                /* <= 16 */
    uint32_t sketchlen;              /* <= MC_MAX_SKETCH */
    uint32_t winlen;                 /* <= MC_MAX_WINLEN */
    uint32_t winstride;
    /* candidate generation (candidate_structs.hpp:113-125) */
    uint32_t max_candidates;         /* K: output stride per query; '-maxcand 0' => caller passes a cap */
    /* table loading */
    uint32_t target_id_bytes;        /* 2 or 4: width of target_id in mc_load_batch values (config.hpp:56-62) */
    uint32_t num_parts;              /* database parts to be loaded (1..255; > 1: every part must be announced with
                                        mc_load_begin before the first mc_load_batch; target ids < 2^24) */
    uint32_t max_locations_per_feature; /* load-time truncation to the first n values (host_hashmap.hpp:454-466); 0 = keep */
    uint32_t remove_overpopulated;   /* load-time: empty buckets with more than n values (host_hashmap.hpp:480-495); 0 = off;
                                        mc_open_database clamps n to the DB's own bucket cap - 1 (mode_query.cpp:69-76) */
    float    max_load_factor;        /* hash table load factor (-max-load-fac, mode_query.cpp:49-55); 0 = default 0.3 */
    /* host batch slots (query_batch ctor, database_query.hpp:192-202) */
    uint32_t num_slots;
    uint32_t slot_max_queries;
    uint32_t slot_max_chars;
    uint32_t copy_allhits;           /* 1: sorted location lists are copied back too (-allhits) */
    int32_t  single_part;            /* mc_open_database: -1 = load every part; p >= 0 = only <name>.cache<p>, as one part
                                        (database::read singlePartId, database.cpp:196-205) -- one part per GPU */
    /* key sharding of ONE part over several GPUs ("Mode K"): only the features f with mc_key_owner(f, count) == index are
     * kept while loading; count <= 1 keeps everything.  A sharded context answers mc_query_device(MC_WANT_ALLHITS) with
     * the PARTIAL location lists of its keys; the union of all shards goes through mc_candidates_from_hits. */
    uint32_t key_shard_index;
    uint32_t key_shard_count;
    /* target-range sharding of ONE part over several GPUs ("Mode T"): mc_open_database keeps only the locations whose target lies in
     * range `index` of `count` CONTIGUOUS target-id ranges (cut where the targets' window counts -- the metadata's -- split into
     * equal shares: target t belongs to range floor(windows_before(t) * count / windows_total)); count <= 1 keeps everything.  The
     * load-time rules above are applied to the file's full bucket first, as on the whole table.  Every range keeps the full target
     * numbering and answers mc_query_* with the unchanged single-table path; the candidates of a read are disjoint between ranges,
     * so mc_merge_part_candidates over the ranges' top lists IN RANGE ORDER equals the whole table's list (mc_partset_open does
     * that when this count is > 1).  Not combinable with key shards or with several parts in one context. */
    uint32_t target_shard_index;
    uint32_t target_shard_count;
} mc_config;

void mc_config_default(mc_config* cfg);

/* lifetime ---------------------------------------------------------------------------------- */
int  mc_create(const mc_config* cfg, mc_ctx** out);
void mc_destroy(mc_ctx* ctx);
const char* mc_last_error(const mc_ctx* ctx);   /* ctx may be NULL: error of the last failed mc_create */

/* table loading: mirrors read_binary(is, featureStore_, partId, progress) (database.cpp:167-179)
 * = hash_multimap::deserialize (hash_multimap.hpp:970-1030), batch structure as in the file
 * (:1037-1082): per batch keys[n] (u32), sizes[n] (u8), values[sum sizes] packed
 * {window_id win (u32); target_id tgt (u16|u32)}. */
int mc_load_begin(mc_ctx* ctx, uint32_t part, uint64_t nkeys, uint64_t nvalues);
/* Optional, before the first mc_load_begin: the number of windows of every target (the reference knows them from its target metadata:
 * database.hpp target_count(), every target's source().windows, taxonomy.hpp:264-280).  When all windows of the database -- plus a gap
 * of 1024 numbers per target -- can be numbered in 32 bits (up to ~480 Gbp at the default window stride, whatever the single targets
 * look like) and the database is a single part, the table stores a location as ONE global window number
 *     gw = first_number(target) + window
 * in 4 bytes instead of the reference's 8-byte {win, tgt} (config.hpp:56-62 / candidate_structs.hpp:44-62): same order, same results,
 * half the bytes per location list in HBM and on the fabric -- a 2 x 10^10-location table takes 86 GB instead of 172.  A location
 * outside its target's announced windows makes mc_load_end fail (MC_ERR_INVALID).  mc_open_database and the builder's tables call
 * this themselves; mc_set_tuning(ctx, "compact_locations", 0) / MC_COMPACT_LOCATIONS=0 keep the 8-byte store. */
int mc_load_target_windows(mc_ctx* ctx, const uint32_t* windows_per_target, uint64_t num_targets);
/* the same announcement from bounds alone: targets 0 .. max_target_id with max_window_id + 1 windows each */
int mc_load_location_range(mc_ctx* ctx, uint32_t max_target_id, uint32_t max_window_id);
/* how the loaded table lies in HBM: layout[0] low 32 bits = bytes per stored location (8, or 4 with the compact store), bit 32 = 1 if the
 * table has its DIRECT-ADDRESS INDEX (2^32 entries of 8 bytes beside the buckets -- the feature is the index: one independent 8-byte
 * load per lookup of the lane path; built at mc_load_end for single-part tables whose buckets take 8 GiB and more where
 * 34 GB + head-room are free; mc_set_tuning "direct_index" 0 / 1 / -1, MC_DIRECT_INDEX), [1] low 32 bits = the gap
 * between two targets' window numbers in the compact form (0 otherwise), high 32 bits = the list alignment in entries (1, or 32 = every
 * list of the compact store begins on a 128-byte line: the loaders switch it on where the padded store stays below 1.5 x the plain one
 * and fits the device; mc_set_tuning "list_align" 0 / 1 / -1, MC_LIST_ALIGN), [2] = number of 64-byte buckets, [3] = entries of the list
 * store (lists of >= 2 with their padding; single locations live in their bucket) */
int mc_table_layout(const mc_ctx* ctx, uint64_t layout[4]);
/* the targets whose locations this context holds: range[0] <= target < range[1] (everything: 0 .. number of targets; a target-range
 * shard, mc_config.target_shard_*: its contiguous range, possibly empty); range[2] = features stored in the table, range[3] = locations */
int mc_target_range(const mc_ctx* ctx, uint64_t range[4]);
int mc_load_batch(mc_ctx* ctx, uint32_t part, const uint32_t* keys, const uint8_t* sizes,
                  const void* values, uint64_t nkeys_in_batch);
int mc_load_end(mc_ctx* ctx, uint32_t part);

/* convenience: database::read (database.cpp:183-242) -- reads <name>.meta (header, sketching,
 * taxonomy -> lineages) and every <name>.cache<p>, creates the context and loads all parts.  Query
 * sketching as adapt_options_to_database (querying.cpp:225-251): kmerlen always the DB's; sketchlen /
 * winlen == 0 => the DB's; winstride == 0 => winlen - kmerlen + 1 (not the DB's stride). */
int mc_open_database(const char* name, const mc_config* cfg, mc_ctx** out);
/* database::read with scope::metadata_only (database.cpp:183-242), what `info` mode needs (mode_info.cpp:55-230): taxa, sources,
 * lineages and the header fields for the mc_db_* calls -- no table, no device; query calls on it fail with MC_ERR_STATE. */
int mc_open_metadata(const char* name, mc_ctx** out);
/* how mc_open_database read the files of a single-part context (reader threads -> pinned slabs -> copy stream -> table kernels; the
 * reference reads every part in its own thread, database.cpp:203-226): stats[0..3] = bytes read, nanoseconds of the load in all, of its
 * index pass (the batches' places in the file), nanoseconds the device feeder waited for the reader threads.  MC_LOAD_THREADS: reader
 * threads (default 8); MC_LOAD_PIPELINE=0: the sequential loader. */
int mc_load_stats(const mc_ctx* ctx, uint64_t stats[4]);

/* target lineage table (ranked_lineages_of_targets, taxonomy.hpp:919-1030; uploaded like
 * copy_target_lineages_to_gpus gpu_hashmap.cu:1383-1396): lin[tgt*21 + rank] = taxon index + 1,
 * 0 = none.  Needed only for lowest_rank > 0 (taxon merging, candidate_generation.hpp:203-228). */
int mc_set_lineages(mc_ctx* ctx, const uint32_t* lin, uint64_t num_targets);

/* database facts after loading: info[0..7] = k, s, w, stride (target sketching), maxLocsPerFeature,
 * targetCount, partCount, locationCount (as stored on device) */
int mc_db_info(const mc_ctx* ctx, uint64_t info[8]);

/* taxonomy block of the .meta file as read by mc_open_database (taxonomy.hpp:702-728): taxa in file
 * order; 'taxon index' everywhere in this ABI = position in this list.  Targets are the taxa with
 * negative ids, id = -(target)-1 (taxonomy.hpp:930). */
int mc_db_num_taxa(const mc_ctx* ctx, uint64_t* n);
int mc_db_taxon(const mc_ctx* ctx, uint64_t index, int64_t* id, int64_t* parent, uint32_t* rank, const char** name);
/* file_source of a taxon (taxonomy.hpp:264-280; meaningful for targets): file name, sequence index in that file, window count */
int mc_db_taxon_source(const mc_ctx* ctx, uint64_t index, const char** filename, uint64_t* file_index, uint64_t* windows);
int mc_db_lineages(const mc_ctx* ctx, const uint32_t** lin, uint64_t* num_targets);

/* host batch slots: query_batch::add_paired_read (query_batch.cuh:85-186), database::query_gpu_async
 * (database.hpp:386-397), host_data.wait_for_results / allhits(i) / top_candidates(i) / clear
 * (query_batch.cuh:212-259); caller code database_query.hpp:87-124 */
int mc_batch_add(mc_ctx* ctx, uint32_t slot, const char* seq1, uint32_t len1, const char* seq2, uint32_t len2,
                 uint32_t max_windows_in_range);
/* many single-end queries at once: read i = seqs[offsets[i] .. offsets[i+1]); maxWindowsInRange is derived per read
 * from the database's window stride (candidate_structs.hpp:143-145) and insert_size_max.  Returns the number of reads
 * added (< n when the slot is full: submit, wait, clear, continue with the rest) or a negative error. */
int64_t mc_batch_add_bulk(mc_ctx* ctx, uint32_t slot, const char* seqs, const uint64_t* offsets, uint64_t n, uint64_t insert_size_max);
/* How the slots reach the device.  With two or more slots of up to 8 192 reads (and top candidates only: copy_allhits = 0) submissions
 * are QUEUED: a slot that finds a pipe free and nothing waiting goes out at once, everything else is taken by dispatcher threads of the
 * library -- whatever is waiting when a pipe comes free goes to the device as ONE batch (slots of the reference's size -- 4 096 reads,
 * options.hpp:229-232 -- are a chain of ~25 dependent steps and two host round trips for 0.1 ms of device work each; database_query.hpp:110-113
 * orders the submissions with a mutex instead; up to five dispatchers: the runtime's four hardware queues); larger slots and MC_SLOT_COALESCE=0: every slot its own batch (MC_SLOT_COALESCE=1: united whatever their size).  stats[0] = 1 if slots are united,
 * [1] = united batches sent so far, [2] = slots they carried, [3] = dispatcher threads. */
int mc_slot_stats(mc_ctx* ctx, uint64_t stats[4]);
/* "" or what the library has noticed about the HIP runtime in this process: the slot paths time their enqueue-only calls, and when these
 * take milliseconds apiece (the runtime's direct dispatch under many submitting threads: 3-4 x slower query phases at 150 Gbp)
 * the text names the remedy -- AMD_DIRECT_DISPATCH=0 in the process' environment before it starts.  Also printed once to stderr. */
const char* mc_runtime_warning(void);
int mc_batch_submit(mc_ctx* ctx, uint32_t slot, int lowest_rank);

typedef struct {
    uint32_t num_queries;
    uint32_t max_candidates;          /* stride of cands */
    const mc_candidate* cands;        /* [num_queries * max_candidates], slot-owned pinned memory */
    const uint64_t* hit_offsets;      /* [num_queries + 1] (NULL unless copy_allhits) */
    const mc_location* hits;          /* sorted (tgt,win) lists, query i = [hit_offsets[i], hit_offsets[i+1]) */
    const uint32_t* hit_counts;       /* [num_queries] number of location hits per query */
} mc_results;

int mc_batch_wait(mc_ctx* ctx, uint32_t slot, mc_results* out);  /* valid until mc_batch_clear(slot) */
int mc_batch_clear(mc_ctx* ctx, uint32_t slot);

/* device-resident entry point (what the slots call after their H2D copy; also used when reads are
 * already in HBM).  All pointers are DEVICE pointers.
 *   seq     : characters; every sequence starts 4-byte aligned; 16 readable slack bytes at the end
 *   qinfo   : [n][4] = {offset1, len1, offset2, len2} (offsets into seq; len2 = 0 for single reads)
 *   max_win : [n] maxWindowsInRange per query (candidate_structs.hpp:143-145), or NULL with
 *             max_win_uniform > 0
 * Results stay on device inside ctx-owned buffers (valid until the next call on this ctx):
 *   cands [n * max_candidates], hit_counts [n] (stride 4 words: {hits, features, found, probe steps}), and -- per
 *   flags -- hit_offsets [n+1] / hits, features.
 * 'stream' is a hipStream_t (NULL = the context's own stream); the call is asynchronous. */
typedef struct {
    const uint8_t*  seq;
    const uint32_t* qinfo;
    const uint32_t* max_win;
    uint32_t        max_win_uniform;
    uint32_t        num_queries;
    uint64_t        num_chars;        /* bytes in seq actually used (for checks) */
} mc_device_batch;

typedef struct {
    const mc_candidate* cands;
    const uint32_t*     hit_counts;
    const uint64_t*     hit_offsets;
    const mc_location*  hits;
    const uint32_t*     features;     /* [total_windows * sketchlen] window sketches, 0xFFFFFFFF padded */
    const uint32_t*     win_offsets;  /* [n+1] first window index of each query */
} mc_device_results;

#define MC_WANT_ALLHITS  1   /* keep the sorted location lists (hit_offsets / hits) */
#define MC_WANT_FEATURES 2   /* keep the window sketches (features) */
#define MC_WANT_PARTIAL_HITS 4 /* keep the location lists (hit_offsets / hits) in ANY order inside a list: what a key-sharded context hands to
                                 the exchange of Mode K (the owner sorts the union); unlike MC_WANT_ALLHITS this keeps the fast lane path */
#define MC_WANT_PARTIAL_NUMBERS 16 /* as MC_WANT_PARTIAL_HITS on a table with the compact location store, the lists left as the 4-byte global
                                 window numbers they are stored as: hit_offsets is filled, hits only for the reads the wave kernels took --
                                 mc_partial_numbers hands the numbers out (no decoding to (target, window) and back: 5.5 -> 0.6 ms per 10^6 reads) */
#define MC_SECOND_PIPE 8     /* run on the context's SECOND pipe (own stream, workspace and result buffers): one more caller thread may have a
                                 call in flight there while another runs on the first pipe, so the kernels of two batches overlap on the
                                 device the way several query_batch objects do in the reference (query_batch.cuh:369-371).  The
                                 results of a pipe stay valid until the next call on the SAME pipe. */
#define MC_DEFER_TAIL 32     /* TWO BATCHES IN FLIGHT FROM ONE CALLER THREAD: the call enqueues the batch's main kernels and returns without any
                                 synchronisation; what the host has to read device counters for (the sorted class of the filtered path, the
                                 segments of the exact wave kernels' leftovers: a few reads per million) is left to mc_query_finish on the
                                 same pipe, which the caller issues AFTER it has enqueued the next batch on the other pipe -- the device
                                 never waits for the host.  out->cands is complete once mc_query_finish has returned (in stream order).
                                 Honoured on the lane path without MC_WANT_* (small batches then skip their look at the work-list counters
                                 too and launch every kernel); otherwise the call does everything at once as without the flag
                                 (mc_query_finish is then a no-op).  A new mc_query_device -- and every other call that uses the first
                                 pipe's workspace: mc_candidates_from_*, mc_partial_numbers, mc_last_batch_stats -- runs a pending tail
                                 first; mc_destroy drops it.  INPUT LIFETIME: the tail reads the batch again (in->seq, in->qinfo,
                                 in->max_win): the caller's device buffers must stay valid and unchanged until mc_query_finish (or the
                                 call that runs the tail in its place) has returned. */
int mc_query_device(mc_ctx* ctx, const mc_device_batch* in, int lowest_rank, int flags,
                    mc_device_results* out, void* stream);
/* the tail of the last mc_query_device(MC_DEFER_TAIL) call on the first (flags = 0) or second (flags = MC_SECOND_PIPE) pipe: waits for
 * that batch's main kernels, then runs its rare classes on the same stream.  The reference gets the same overlap from several
 * query_batch objects, one stream each (query_batch.cuh:369-371, database_query.hpp:110-113). */
int mc_query_finish(mc_ctx* ctx, int flags);
/* waits for ONE pipe's stream (flags = 0: the first, MC_SECOND_PIPE: the second) -- what a two-pipe caller uses instead of mc_synchronize
 * (which waits for both) before it hands a finished batch's results to somebody else while the other pipe's batch is still running */
int mc_query_wait(mc_ctx* ctx, int flags);
int mc_synchronize(mc_ctx* ctx);

/* Mode P (one database part per GPU) and part groups.  Per-part candidate lists of the same reads (DEVICE pointers, [n][max_candidates]
 * each, in part order; target ids as mc_open_database with cfg.single_part leaves them: the database's own) -> one list per read, as if
 * the parts had been queried one after the other with one candidate list: host_hashmap.hpp:695-723 concatenates the parts' sorted
 * location lists, candidate_generation.hpp:172-231 inserts their candidates in that order (ties keep arrival order; lowest_rank > 0: one
 * entry per taxon, this context's lineages).  max_candidates <= 4.  out may be lists[0].  Asynchronous on 'stream'.
 * Replaces what the reference does across its GPUs for a partitioned database (gpu_hashmap.cu:1255-1290, query_batch.cu:464-527) and
 * `metacache merge` of per-part result files (mode_merge.cpp:247-296). */
int mc_merge_part_candidates(mc_ctx* ctx, const mc_candidate* const* lists, uint32_t num_lists, uint32_t num_queries, int lowest_rank,
                             mc_candidate* out, void* stream);

/* A partitioned database queried part group by part group (docs/partitioning.md:116-153, for databases beyond the node's HBM), the
 * parts of a group spread over GPUs (options.cpp:1155-1163 lists the reference's multi-GPU switches): `resident_parts` parts are in
 * HBM at a time -- one context each, part i of a group on devices[i % num_devices] -- and the NEXT group is loaded by a background
 * thread while the reads run against this one.  Per batch the reads are dealt out to the devices as owners; every device sends every
 * owner its parts' top lists of that owner's reads (one grouped ncclSend / ncclRecv round: RCCL, one communicator rank per device,
 * ncclCommInitAll) and every owner merges its reads' lists in part order (mc_merge_part_candidates), together with the list the earlier
 * groups left, and copies its share to the host (the reference forwards running top candidates GPU -> GPU, query_batch.cu:638-652).  Two
 * batches are in flight; mc_partset_classify_resident may be called from several threads at once.  devices == NULL: cfg->device alone.  cfg: as for mc_open_database (slot_max_queries / slot_max_chars = batch size).
 * cfg->target_shard_count > 1: the "parts" are the contiguous target ranges of ONE part file (cfg->single_part, or part 0), cut at load
 * (mc_config.target_shard_*); everything else -- groups, gather, merge in range order -- is the same. */
typedef struct mc_partset mc_partset;
int  mc_partset_open(const char* name, const mc_config* cfg, uint32_t resident_parts, const int32_t* devices, uint32_t num_devices, mc_partset** out);
void mc_partset_close(mc_partset* ps);
/* info[0..5] = parts, resident parts, groups, devices, nanoseconds the loader thread spent loading, nanoseconds the queries waited for it */
int  mc_partset_info(const mc_partset* ps, uint64_t info[6]);
/* all n reads (read i = seqs + offs[i] .. offs[i + 1]; seqs2 / offs2: the mates, NULL for single reads) against every part;
 * out[n][max_candidates] in HOST memory; insert_max as -insertsize (maxWindowsInRange, candidate_structs.hpp:143-145) */
int  mc_partset_classify(mc_partset* ps, const char* seqs, const uint64_t* offs, const char* seqs2, const uint64_t* offs2, uint64_t n,
                         int lowest_rank, uint64_t insert_max, mc_candidate* out);
/* The same in the caller's hands, for reads that STREAM (mcq -resident-parts): mc_partset_select_group makes part group g resident (groups in
 * order 0, 1, ...: group g + 1 loads behind group g's queries, every part by its own reader threads, one loading thread per device --
 * database.cpp:203-226 reads every part in a thread of its own); mc_partset_classify_resident takes a batch of reads through the parts of
 * the resident group only and merges their lists behind the list the earlier groups left in inout (has_prior != 0) -- the caller keeps
 * 16 bytes x max_candidates per read between the groups, nothing else.  mc_partset_load_bytes: bytes of .cache files the group loads read. */
int  mc_partset_select_group(mc_partset* ps, uint32_t group);
int  mc_partset_classify_resident(mc_partset* ps, const char* seqs, const uint64_t* offs, const char* seqs2, const uint64_t* offs2, uint64_t n,
                                  int lowest_rank, uint64_t insert_max, int has_prior, mc_candidate* inout);
int  mc_partset_load_bytes(const mc_partset* ps, uint64_t* bytes);
const char* mc_partset_last_error(const mc_partset* ps);

/* Mode K.  Owner shard of a feature (independent of the table's own bucket hash). */
uint32_t mc_key_owner(uint32_t feature, uint32_t shard_count);
/* rows 8-10 on location lists that are already gathered (DEVICE pointers; query i = hits[hit_offsets[i] .. hit_offsets[i+1]),
 * any order inside a list, e.g. the concatenated partial lists of all key shards): sort by (tgt, win), best window range per
 * target, top candidates -- query_handler.hpp:75-101 + candidate_generation.hpp:47-231 on a single-part list.
 * Results as mc_query_device (out->cands; out->hits / hit_offsets = the sorted lists).  Asynchronous on 'stream'. */
typedef struct {
    const mc_location* hits;
    const uint64_t*    hit_offsets;   /* [n + 1] */
    const uint32_t*    max_win;       /* [n] or NULL with max_win_uniform > 0 */
    uint32_t           max_win_uniform;
    uint32_t           num_queries;
} mc_device_hits;
int mc_candidates_from_hits(mc_ctx* ctx, const mc_device_hits* in, int lowest_rank, mc_device_results* out, void* stream);
/* Mode K owner side in ONE call, everything on the device: the partial lists of num_sources key shards for this rank's num_queries
 * reads -- counts[s * num_queries + i] locations of read i from source s; each source's locations back to back in read order, the
 * sources' blocks back to back in hits (exactly what an all-to-all-v of the shards' sorted partial lists delivers; total_hits = its
 * receive size, known to the host from the split sizes) -- are concatenated per read and go through rows 8-10: united lists of more
 * than 256 locations through the target filter and the (target, window) counting of the replicated mode (the union buffer standing
 * in for the table's location store), the others through the sort of mc_candidates_from_hits.  out->hits / hit_offsets = the united
 * lists (sorted only where the sort ran).  Replaces the reference's per-part forwarding chain (query_batch.cu:464-527, :638-652). */
typedef struct {
    const uint32_t*    counts;        /* [num_sources * num_queries] */
    const mc_location* hits;          /* [total_hits] */
    uint64_t           total_hits;
    const uint32_t*    max_win;       /* [num_queries] or NULL with max_win_uniform > 0 */
    uint32_t           max_win_uniform;
    uint32_t           num_queries;
    uint32_t           num_sources;
} mc_device_partial_hits;
int mc_candidates_from_partial_hits(mc_ctx* ctx, const mc_device_partial_hits* in, int lowest_rank, mc_device_results* out, void* stream);
/* Mode K with 4-BYTE locations on the wire.  What travels between the key shards is the compact location store's own form of a
 * location, the global window number gw = gwBase[target] + window (every context of one database numbers the windows of ALL targets the
 * same way: the numbering comes from the targets' window counts in the metadata).  Needs a single-part database whose windows can be
 * numbered in 32 bits (mc_load_target_windows; MC_ERR_UNSUPPORTED otherwise: use the 8-byte calls above).
 *
 * Shard side, after mc_query_device(MC_WANT_PARTIAL_HITS) on a key-sharded context (such a call looks up only the features the shard
 * owns): res's partial lists as numbers, back to back in read order, and the per-read counts -- the piece for the owner of reads
 * [lo, hi) is the contiguous range numbers[cut(lo) .. cut(hi)).  cut_offsets[i] (HOST) = first number of read cut_queries[i]
 * (cut_queries[i] <= num_queries): the split sizes of the all-to-all-v, the one host round trip of the exchange.  out's pointers
 * are ctx-owned device memory, valid until the next mc_partial_numbers call; filled asynchronously on 'stream'.
 * Replaces query_batch.cu:464-527 (the reference forwards the sketches from GPU to GPU and accumulates per-part results). */
typedef struct {
    const uint32_t* counts;       /* device [num_queries] */
    const uint32_t* numbers;      /* device [total] */
    uint64_t        total;
} mc_device_partial_numbers;
int mc_partial_numbers(mc_ctx* ctx, const mc_device_results* res, uint32_t num_queries, const uint32_t* cut_queries, uint32_t num_cuts,
                       uint64_t* cut_offsets, mc_device_partial_numbers* out, void* stream);
/* Owner side in ONE call: rows 8-10 on the pieces num_sources (<= 64) key shards sent for this rank's num_queries reads --
 * counts[s * num_queries + i] numbers of read i from source s, each source's numbers back to back in read order, source s' block
 * at numbers[source_offsets[s] .. source_offsets[s + 1]) (source_offsets on the HOST: the receive displacements).  No union copy:
 * the receive buffer stands in for the table's location store, every source's piece is one list of the read; lists of more than
 * 64 locations take the filtered path of the replicated mode (gw_filter / gw_count / sorted lists), the rest and what that path
 * hands back are decoded to (target, window) and sorted.  'numbers' must stay untouched until the results have been copied out and
 * must be readable 16 bytes past its end.  Results as mc_query_device (out->cands).
 * Replaces query_batch.cu:638-652 + the candidate generation of gpu_hashmap.cu:1255-1290 across the reference's GPUs. */
typedef struct {
    const uint32_t* counts;          /* device [num_sources * num_queries] */
    const uint32_t* numbers;         /* device */
    const uint64_t* source_offsets;  /* HOST [num_sources + 1] */
    const uint32_t* max_win;         /* device [num_queries] or NULL with max_win_uniform > 0 */
    uint32_t        max_win_uniform;
    uint32_t        num_queries;
    uint32_t        num_sources;
} mc_device_partial_numbers_in;
int mc_candidates_from_partial_numbers(mc_ctx* ctx, const mc_device_partial_numbers_in* in, int lowest_rank, mc_device_results* out, void* stream);
/* the same on the context's first (flags = 0) or second (flags = MC_SECOND_PIPE) pipe: a caller that keeps two batches in flight (keyset.cpp's
 * lanes) runs a batch's shard side -- mc_query_device(MC_WANT_PARTIAL_NUMBERS | pipe), mc_partial_numbers -- and its owner side on the same pipe */
int mc_candidates_from_partial_numbers_on(mc_ctx* ctx, const mc_device_partial_numbers_in* in, int lowest_rank, int flags, mc_device_results* out, void* stream);
/* what mc_candidates_from_partial_numbers has done on this context so far: stats[0..3] = reads, reads whose pieces took the filtered path,
 * numbers received, locations decoded for the sort (short lists + what the filtered path handed back) */
int mc_owner_stats(const mc_ctx* ctx, uint64_t stats[4]);

/* ONE database key-sharded over the GPUs of the node, driven from C++ (what `mcq query -shard keys -gpus a,b,...` runs): shard s of
 * num_shards on devices[s % num_devices], one context each (mc_open_database with key_shard_index / key_shard_count).  Per batch
 * every shard looks its own features up for ALL reads (mc_query_device(MC_WANT_PARTIAL_HITS)), the partial lists travel as 4-byte
 * numbers to the shard that owns the read (contiguous read shards; grouped ncclSend / ncclRecv = all-to-all-v over RCCL/xGMI between
 * devices, device-to-device copies between shards of one device), and the owner runs rows 8-10 (mc_candidates_from_partial_numbers).
 * RCCL is used when num_shards == num_devices > 1 (or MC_KEYSET_RCCL=1 with one shard: the same calls with a single rank);
 * several shards on one device are what a single-GPU box can test.  num_shards == 0: one per device.
 * Reference: gpu_hashmap.cu:1255-1290, query_batch.cu:464-527 (its parts-over-GPUs query), options.cpp:1155-1163. */
typedef struct mc_keyset mc_keyset;
int  mc_keyset_open(const char* name, const mc_config* cfg, uint32_t num_shards, const int32_t* devices, uint32_t num_devices, mc_keyset** out);
void mc_keyset_close(mc_keyset* ks);
/* info[0..7] = shards, devices, 1 if the exchange runs over RCCL, locations of all shards, numbers sent through the exchange so far, batches,
 * reads whose pieces took the filtered path on their owner, locations decoded for the sort (mc_owner_stats summed over the shards) */
int  mc_keyset_info(const mc_keyset* ks, uint64_t info[8]);
/* as mc_partset_classify: out[n][max_candidates] in HOST memory */
int  mc_keyset_classify(mc_keyset* ks, const char* seqs, const uint64_t* offs, const char* seqs2, const uint64_t* offs2, uint64_t n,
                        int lowest_rank, uint64_t insert_max, mc_candidate* out);
const char* mc_keyset_last_error(const mc_keyset* ks);

/* copies out of the ctx-owned result buffers, asynchronous on the context's stream
 * (kind: 0 = device -> device, 1 = device -> host); mc_synchronize waits for both pipes' streams */
int mc_copy_results(mc_ctx* ctx, void* dst, const void* src, uint64_t bytes, int kind);
/* the same on the caller's stream (the one its mc_query_device call ran on; NULL = the context's, or with kind | MC_SECOND_PIPE the
 * second pipe's own stream: results of a mc_query_device(MC_SECOND_PIPE) call that was given no stream).
 * kind 2 = host -> device: a caller whose reads lie in pinned host memory uploads batch i + 1 on the pipe it will run on while batch i's
 * kernels run on the other pipe (cudaMemcpyAsync of query_batch's host buffers, query_batch.cu:330-360) */
int mc_copy_results_on(mc_ctx* ctx, void* dst, const void* src, uint64_t bytes, int kind, void* stream);

/* tuning / test hook (not needed for normal use): the switches the MC_BIG_MIN / MC_QUAD_LOOKUP / MC_NO_LANE_PATH environment variables
 * set at mc_create, on a live context with no batch in flight.  names: "big_min" (location lists longer than this are filtered by
 * target before they are counted, big_filter_kernel), "quad_lookup" (-1 by table size, 0 / 1), "lane_path" (0 / 1), "compact_locations" (0 / 1, before mc_load_begin),
 * "filter_bpc" / "count_bpc" (blocks per CU of the filter kernels' / the first counting instance's persistent grids, 0 = default; this context only),
 * "gw_fuse" (counting inside the filter kernel: 1 = default, 0 = the two kernels apart, 5 / 6 = the fused kernel's five- / six-waves-per-SIMD instances; the default runs at seven),
 * "gw_big_h" (reads beyond this many locations take the fine-block instance of the stream filter; default 32 768, 0 = none),
 * "lane_fusion" (sketching + lookups of the lane path in one kernel: -1 = on tables beyond 1 GiB (default), 0 / 1 = never / always),
 * "direct_index" (-1 by table size, 0 / 1; on a loaded table the index is built or dropped at once: mc_table_layout),
 * "list_align" (before the table is loaded: the compact store's lists on 128-byte lines of their own, mc_table_layout).
 * Every value of every switch gives the same results (tests/test_gpu_variants.py and the variant loops of test_gpu_scale.py /
 * test_gpu_reference_midscale.py run them against the goldens, the oracle and the reference). */
int mc_set_tuning(mc_ctx* ctx, const char* name, int64_t value);

/* per-kernel timing with HIP events on the launching stream (for bench.py's roofline block).
 * names: "plan", "sketch_lane", "chunk_sketch", "chunk_probe", "probe_cands", "mid_cands_64", "mid_cands_128", "mid_cands_256",
 * "hash_cands_256", "hash_cands_512", "hash_cands_1024", the filtered path -- compact location store: "gw_filter_count" (gw_filter_count_kernel; "gw_filter" with
 * the tuning switch "gw_fuse" 0), "gw_filter2", "gw_compact" (+ the ordering of the stream filter's reads), "gw_filter_stream_fine" (the sixteen-wave instance), "gw_filter_stream" (+ the second gw_compact), "gw_count" (gw_count_kernel<9>), "gw_count_512" (<10>), "gw_count_1024" (<11>); 8-byte store:
 * "big_filter", "big_filter_2", "big_count", "big_count_2" --, "gw_sort", "gw_sorted_cands", "query_wave", "scan", "sort_candidates";
 * Mode K: "mask_features", "gather_lists", "pack_numbers", "owner_entries", "decode_union"; "sketch_probe" (sketch_probe_lane_kernel: instead of "sketch_lane" + "probe_cands" where the two are one kernel, see "lane_fusion"); "cands_from_hits" (mc_candidates_from_hits).  Returns accumulated milliseconds and launch counts since the last reset. */
int mc_timing_enable(mc_ctx* ctx, int on);
int mc_timing_reset(mc_ctx* ctx);
int mc_timing_get(mc_ctx* ctx, const char* kernel, double* total_ms, uint64_t* launches);

/* workload statistics of the LAST mc_query_device call (device reductions, synchronises):
 * stats[0] = total windows, [1] = valid features probed (F), [2] = locations returned (H),
 * [3] = features found in table, [4] = probe steps (bucket groups read) */
int mc_last_batch_stats(mc_ctx* ctx, uint64_t stats[8]);

/* ---- minimal database builder ------------------------------------------
 * Needed so that synthetic databases can be produced on the GPU box itself; it re-uses the
 * parity-proven sketch kernel.  Mirrors database::add_target (database.cpp:34-81) +
 * host_hashmap::add_target (host_hashmap.hpp:570-589) for a SINGLE-threaded build: target ids in
 * call order, window ids = running index of windows with >= k characters, buckets hold the first
 * max_locations_per_feature locations in (target, window) order (host_hashmap.hpp:593-605), and
 * database::write (database.cpp:247-325) for the file format.  cfg fields used: device, kmerlen,
 * sketchlen, winlen, winstride, target_id_bytes, max_locations_per_feature (0 => 254), remove_overpopulated (!= 0 => features
 * that reached the limit are dropped from the files and the table: -remove-overpopulated-features, building.cpp:516-534),
 * key_shard_index / key_shard_count (> 1: only this shard's features are kept; see mc_build_finish_shards). */
typedef struct mc_builder mc_builder;
typedef struct {
    int64_t  id;
    int64_t  parent;
    uint32_t rank;          /* taxonomy::rank as integer (taxonomy.hpp:68-91), 21 = none */
    const char* name;
} mc_taxon_rec;

int  mc_build_begin(const mc_config* cfg, mc_builder** out);
int  mc_build_add_target(mc_builder* b, const char* seq, uint64_t len, const char* name, int64_t parent_taxid,
                         const char* source_filename);
/* the same with the position of the sequence inside its file (taxon::file_source::index, building.cpp:414-416) */
int  mc_build_add_target_src(mc_builder* b, const char* seq, uint64_t len, const char* name, int64_t parent_taxid,
                             const char* source_filename, uint64_t source_index);
/* the same for a sequence that already lies in DEVICE memory (start 4-byte aligned, 16 readable bytes behind its last character):
 * nothing is staged or copied, the sketch kernels read it in place.  The memory must stay untouched until mc_build_flush or
 * mc_build_finish returns.  (The reference's GPU build takes host sequences only, gpu_hashmap.cu:1024-1120; collections that
 * are generated or decoded on the device -- bench.py's RefSeq-scale synthetic database -- go in without a PCIe round trip.) */
int  mc_build_add_target_device(mc_builder* b, const void* dseq, uint64_t len, const char* name, int64_t parent_taxid,
                                const char* source_filename, uint64_t source_index);
/* sketches everything staged so far (sources of mc_build_add_target_device calls may be reused afterwards) */
int  mc_build_flush(mc_builder* b);
/* allocates room for this many (feature, location) pairs up front (a builder that grows step by step holds two copies while growing) */
int  mc_build_reserve(mc_builder* b, uint64_t pairs);
/* re-ranks a target after it was added (try_to_rank_unranked_targets, building.cpp:196-232) */
int  mc_build_set_parent(mc_builder* b, uint64_t target, int64_t parent_taxid);
/* modify mode (main_mode_modify, mode_build.cpp:74-88: an existing database is read, then added to): the database's targets
 * (taxon::file_source fields from mc_db_taxon / mc_db_taxon_source) and the batches of its .cache file (hash_multimap.hpp:1037-1082:
 * keys, bucket sizes, packed {u32 window, target id of target_bytes} values) go into a fresh builder BEFORE the first new target;
 * their lists keep their order and stand in front of what is sketched afterwards, as in the reference's table after reading. */
int  mc_build_add_existing_target(mc_builder* b, const char* name, int64_t parent_taxid, const char* source_filename,
                                  uint64_t source_index, uint64_t windows);
int  mc_build_add_locations(mc_builder* b, const uint32_t* keys, const uint8_t* sizes, const void* values, uint64_t num_keys,
                            uint32_t target_bytes);
/* after mc_build_finish: features and locations the builder holds (database::feature_count / location_count, database.hpp:420-440) */
int  mc_build_counts(const mc_builder* b, uint64_t* keys, uint64_t* values);
/* After mc_build_finish: drops every feature whose locations lie in more than max_ambig (0 => 1) different taxa on one rank
 * (-remove-ambig-features <rank> -max-ambig-per-feature <n>: database.hpp:259-270, host_hashmap.hpp:499-540, called from
 * post_process_features, building.cpp:550-566).  ancestor_of_target[t] = any id of target t's ancestor on that rank, 0 = none
 * (all targets without one count as ONE taxon, as the reference's null pointer does); rank 'sequence': t + 1.  Key-sharded
 * builders: call it on every shard.  *removed (may be NULL) = number of features dropped. */
int  mc_build_remove_ambiguous(mc_builder* b, const uint32_t* ancestor_of_target, uint64_t num_targets, uint32_t max_ambig,
                               uint64_t* removed);
/* number of windows of a target (taxon::file_source::windows, database.cpp:64) */
int  mc_build_target_windows(const mc_builder* b, uint64_t target, uint64_t* windows);
/* sorts + bucketises everything added so far; if out_ctx != NULL also loads the table into a fresh
 * query context (see mc_build_set_query_config). */
int  mc_build_finish(mc_builder* b, mc_ctx** out_ctx);
/* One query table from n builders that were given the SAME targets and the key shards 0 .. n-1 of n (cfg.key_shard_index / _count:
 * a builder keeps only the features of its shard, as a Mode K context does): tables beyond the 2^32 (feature, location) pairs one
 * device sort takes are built shard after shard.  n == 1: the loading half of mc_build_finish. */
int  mc_build_finish_shards(mc_builder** builders, uint32_t n, mc_ctx** out_ctx);
/* Streaming form of mc_build_finish_shards for tables that do not fit next to all their builders (RefSeq scale: 2 x 10^10 locations):
 *   mc_build_table_begin  creates the query context and sizes its table for expect_keys features / expect_values locations
 *                         (0 = estimate from the FINISHED builder b: its counts times its key_shard_count, plus a margin);
 *   mc_build_table_add    inserts one finished builder -- a whole one or one key shard, same targets -- from its device arrays;
 *                         the builder can be freed right after;
 *   mc_build_table_end    closes the load (mc_load_end).  More keys or locations than announced fail with MC_ERR_INVALID.
 * Between _begin and _end (and between mc_partset_open and mc_partset_close) device blocks of 64 MB and more that the library frees are kept
 * for its next allocation of that size instead of going back to the device -- hipMalloc hands out scrubbed memory at ~18 GB/s, which was
 * most of a build's and of a part group's time --: at most MC_DEVCACHE_GB (environment, default 64) gigabytes, released by the closing
 * call or when an allocation cannot be served. */
int  mc_build_table_begin(mc_builder* b, uint64_t expect_keys, uint64_t expect_values, mc_ctx** out_ctx);
int  mc_build_table_add(mc_ctx* ctx, mc_builder* shard);
int  mc_build_table_end(mc_ctx* ctx);
/* fields of the query context mc_build_finish creates (max_candidates, slots, copy_allhits, load factor) */
int  mc_build_set_query_config(mc_builder* b, const mc_config* qcfg);
/* writes <name>.meta and <name>.cache0 in the reference's format (after mc_build_finish) */
int  mc_build_write(mc_builder* b, const char* name, const mc_taxon_rec* taxa, uint64_t ntaxa);
/* the same for the builders of one key-sharded set (see mc_build_finish_shards): one complete database */
int  mc_build_write_shards(mc_builder** builders, uint32_t n, const char* name, const mc_taxon_rec* taxa, uint64_t ntaxa);
/* the same shard by shard (a RefSeq-scale set of builders does not exist at once): begin writes <name>.meta (targets of b) and opens
 * <name>.cache0, add appends one finished builder (key shards 0 .. n-1 in order), end patches the totals into the header and closes;
 * any failure removes both files. */
typedef struct mc_db_writer mc_db_writer;
int  mc_build_write_begin(mc_builder* b, const char* name, const mc_taxon_rec* taxa, uint64_t ntaxa, mc_db_writer** out);
int  mc_build_write_add(mc_db_writer* w, mc_builder* shard);
int  mc_build_write_end(mc_db_writer* w);
void mc_build_free(mc_builder* b);
const char* mc_build_last_error(const mc_builder* b);

#ifdef __cplusplus
}
#endif
#endif /* METACACHE_AMD_H_ */
