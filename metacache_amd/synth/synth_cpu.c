/* metacache_amd/synth/synth_cpu.c -- host side of the synthetic workload (synth_spec.h): the same targets and reads the GPU
 * generator writes, for the CPU checkers (oracle) and for tests without a GPU.  WORKLOAD GENERATION ONLY. */
#include "synth_spec.h"

#include <string.h>

/* ASCII bases [first, first + n) of one target */
void mcs_cpu_target(const syn_target* t, uint32_t first, uint32_t n, uint8_t* dst)
{
    for (uint32_t i = 0; i < n; ++i) dst[i] = syn_ascii(syn_code(t, first + i));
}

/* callback form used by the oracle's database restatement: user = syn_target array, writes the whole target */
void mcs_cpu_target_cb(void* user, uint32_t target, char* dst)
{
    const syn_target* t = (const syn_target*)user + target;
    mcs_cpu_target(t, 0, t->length, (uint8_t*)dst);
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
