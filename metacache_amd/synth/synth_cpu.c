/* metacache_amd/synth/synth_cpu.c -- host side of the synthetic workload (synth_spec.h): the same targets and reads the GPU
 * generator writes, for the CPU checkers (oracle) and for tests without a GPU.  WORKLOAD GENERATION ONLY. */
#include "synth_spec.h"

#include <stdlib.h>
#include <string.h>

/* ASCII bases [first, first + n) of one target */
void mcs_cpu_target(const syn_target* t, uint32_t first, uint32_t n, uint8_t* dst)
{
    for (uint32_t i = 0; i < n; ++i) dst[i] = syn_ascii(syn_code(t, first + i));
}

/* callback form used by the checker's database restatement: user = syn_target array, writes the whole target.  Targets of one
 * genus / species share their ancestor: a thread that is handed consecutive targets keeps the 2-bit codes of the last genus
 * ancestor and the last species it produced (same function values as syn_code, two hashes per base less for most targets). */
void mcs_cpu_target_cb(void* user, uint32_t target, char* dst)
{
    static __thread uint64_t gseed = 0, sseed = 0;
    static __thread uint32_t glen = 0, sthr = 0, cap = 0;
    static __thread uint8_t *gcode = 0, *scode = 0;
    const syn_target* t = (const syn_target*)user + target;
    const uint32_t n = t->length;
    if (n > cap) { free(gcode); free(scode); cap = n; gcode = (uint8_t*)malloc(cap); scode = (uint8_t*)malloc(cap); gseed = sseed = 0; }
    if (t->genus_seed != gseed || n != glen) {
        for (uint32_t p = 0; p < n; ++p) gcode[p] = (uint8_t)(syn_h(t->genus_seed, p) & 3u);
        gseed = t->genus_seed; glen = n; sseed = 0;
    }
    if (t->species_seed != sseed || t->thr_species != sthr) {
        for (uint32_t p = 0; p < n; ++p) {
            uint32_t c = gcode[p];
            const uint32_t hs = syn_h(t->species_seed, p);
            if (hs < t->thr_species) c = (c + 1u + hs % 3u) & 3u;
            scode[p] = (uint8_t)c;
        }
        sseed = t->species_seed; sthr = t->thr_species;
    }
    for (uint32_t p = 0; p < n; ++p) {
        uint32_t c = scode[p];
        const uint32_t ht = syn_h(t->strain_seed, p);
        if (ht < t->thr_strain) c = (c + 1u + ht % 3u) & 3u;
        dst[p] = (char)syn_ascii(c);
    }
}

/* reads [first, first + n) into rows of P->row_bytes (zero padded); paired: mate 2 rows follow at dst2 */
void mcs_cpu_reads(const syn_read_params* P, const syn_target* targets, uint64_t first, uint64_t n, uint8_t* dst, uint8_t* dst2)
{
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t r = first + i;
        const syn_read_origin o = syn_origin(P, targets, r);
        const syn_target* t = &targets[o.target];
        for (uint32_t m = 0; m < (P->paired ? 2u : 1u); ++m) {
            uint8_t* row = (m ? dst2 : dst) + i * P->row_bytes;
            for (uint32_t j = 0; j < P->read_len; ++j) row[j] = syn_read_char(P, t, o, r, m, j);
            memset(row + P->read_len, 0, P->row_bytes - P->read_len);
        }
    }
}

void mcs_cpu_origins(const syn_read_params* P, const syn_target* targets, uint64_t first, uint64_t n, uint32_t* out4)
{
    for (uint64_t i = 0; i < n; ++i) {
        const syn_read_origin o = syn_origin(P, targets, first + i);
        out4[4 * i] = o.target; out4[4 * i + 1] = o.start; out4[4 * i + 2] = o.reverse; out4[4 * i + 3] = o.frag;
    }
}
