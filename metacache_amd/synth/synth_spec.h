/* metacache_amd/synth/synth_spec.h -- the synthetic workload of BASELINE.json configs[2..4] as a pure function.
 *
 * WORKLOAD GENERATION ONLY (bench.py, tests, tools): nothing here is on the measured path or part of the product library.
 *
 * There is no network, so a "RefSeq-scale" database has to be generated where it is used (SURVEY.md §8d "Config 3"): 150 Gbp of
 * target sequences do not fit next to the table they are turned into, and the CPU checker needs the same sequences on the host.
 * Therefore every base of every target is a FUNCTION of (target, position) -- counter-based hashing, no state -- so that
 *   * the GPU generator (synth.hip) writes any stretch of any target into HBM when the builder asks for it, once per key-shard pass,
 *   * the CPU generator (synth_cpu.c) gives the oracle the very same characters,
 *   * reads are drawn by evaluating the function at (target, start .. start + L) -- no genome has to be resident.
 *
 * Phylogeny (genus -> species -> strain, substitutions only): a genus has an ancestor sequence; a species is the ancestor with a
 * fraction thrSpecies / 2^32 of its positions substituted; a strain is its species with a further thrStrain / 2^32 substituted.
 * Strains of a species therefore share most 16-mers (heavy-tailed feature buckets, ties between strains), species of a genus
 * share a few, genera nothing -- the structure that makes real reference collections hard.
 *
 * Plain C99 that also compiles as HIP device code (SYN_HD).
 */
#ifndef MC_SYNTH_SPEC_H_
#define MC_SYNTH_SPEC_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define SYN_HD __host__ __device__ static inline
#else
#define SYN_HD static inline
#endif

typedef struct {
    uint64_t genus_seed, species_seed, strain_seed;
    uint32_t thr_species, thr_strain;   /* substitution probability * 2^32 */
    uint32_t length;                    /* bases */
    uint32_t pad_;
} syn_target;                           /* 40 bytes */

typedef struct {
    uint64_t seed;
    uint32_t read_len;      /* bases per read (mate) */
    uint32_t row_bytes;     /* bytes per output row (>= read_len, multiple of 4; the rest is zero) */
    uint32_t thr_sub;       /* per-base substitution probability * 2^32 */
    uint32_t thr_n;         /* per-base 'N' probability * 2^32 */
    uint32_t paired;        /* 0: single reads.  1: row i = mate 1, row n + i = mate 2 (reverse strand of the fragment's other end) */
    uint32_t frag_min, frag_max;   /* paired: fragment length uniform in [frag_min, frag_max] */
    uint32_t num_targets;
} syn_read_params;

SYN_HD uint32_t syn_fmix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}

/* counter-based hash: 32 well-mixed bits for (seed, position) */
SYN_HD uint32_t syn_h(uint64_t seed, uint32_t p)
{
    return syn_fmix32(syn_fmix32(p + (uint32_t)seed) ^ (uint32_t)(seed >> 32));
}

/* 2-bit code (A0 C1 G2 T3) of base p of a target */
SYN_HD uint32_t syn_code(const syn_target* t, uint32_t p)
{
    uint32_t c = syn_h(t->genus_seed, p) & 3u;
    const uint32_t hs = syn_h(t->species_seed, p);
    if (hs < t->thr_species) c = (c + 1u + hs % 3u) & 3u;
    const uint32_t ht = syn_h(t->strain_seed, p);
    if (ht < t->thr_strain) c = (c + 1u + ht % 3u) & 3u;
    return c;
}

SYN_HD uint8_t syn_ascii(uint32_t code) { return (uint8_t)("ACGT"[code & 3u]); }

/* where read r comes from: target, first base, strand (1 = reverse complement), fragment length (paired) */
typedef struct { uint32_t target, start, reverse, frag; } syn_read_origin;

SYN_HD syn_read_origin syn_origin(const syn_read_params* P, const syn_target* targets, uint64_t r)
{
    const uint64_t rs = P->seed ^ (r * 0x9E3779B97F4A7C15ull);
    syn_read_origin o;
    o.target = (uint32_t)(((uint64_t)syn_h(rs, 0) * P->num_targets) >> 32);
    uint32_t span = P->read_len;
    o.frag = 0;
    if (P->paired) {
        o.frag = P->frag_min + (uint32_t)(((uint64_t)syn_h(rs, 3) * (P->frag_max - P->frag_min + 1u)) >> 32);
        span = o.frag;
    }
    const uint32_t L = targets[o.target].length;
    o.start = L > span ? (uint32_t)(((uint64_t)syn_h(rs, 1) * (L - span + 1u)) >> 32) : 0u;
    o.reverse = syn_h(rs, 2) & 1u;
    return o;
}

/* character j of mate m (0 / 1) of read r, whose origin o = syn_origin(P, targets, r) lies on target t = &targets[o.target] */
SYN_HD uint8_t syn_read_char(const syn_read_params* P, const syn_target* t, syn_read_origin o, uint64_t r, uint32_t mate, uint32_t j)
{
    const uint32_t RL = P->read_len;
    /* the fragment [start, start + span) in forward orientation; reverse = the read pair is drawn from the other strand.
     * mate 1 reads the fragment's 5' end forwards, mate 2 its 3' end backwards (reverse complement), as sequencers do. */
    const uint32_t span = P->paired ? o.frag : RL;
    uint32_t fpos;            /* position inside the fragment, read direction given by 'rc' */
    uint32_t rc = mate;       /* mate 2 is reverse-complemented relative to the fragment */
    fpos = mate ? span - 1u - j : j;
    if (o.reverse) { fpos = span - 1u - fpos; rc ^= 1u; }
    const uint32_t gp = o.start + fpos;
    if (gp >= t->length) return (uint8_t)'N';
    uint32_t c = syn_code(t, gp);
    if (rc) c = 3u - c;
    const uint64_t es = (P->seed ^ (r * 0xD1B54A32D192ED03ull)) + mate;
    const uint32_t e = syn_h(es, j);
    if (e < P->thr_sub) c = (c + 1u + e % 3u) & 3u;
    if (syn_h(es ^ 0x5851F42D4C957F2Dull, j) < P->thr_n) return (uint8_t)'N';
    return syn_ascii(c);
}

#endif /* MC_SYNTH_SPEC_H_ */
