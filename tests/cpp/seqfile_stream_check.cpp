#include "mcq_common.h"
using namespace mcq;
static std::string dump(SeqFile& f) { std::string out, scratch; View h, s; for (size_t i = 0; i < f.records(); ++i) { f.record(i, h, s, scratch); out.append(h.p, h.n); out += '\t'; out.append(s.p, s.n); out += '\n'; } return out; }
int main(int argc, char** argv) {
    int bad = 0;
    for (int a = 1; a < argc; ++a) {
        SeqFile e(argv[a]); e.index(4);
        const std::string want = dump(e);
        for (const char* chunk : {"64", "500", "3000", "100000"}) {
            setenv("MCQ_STREAM_MIN", "0", 1); setenv("MCQ_STREAM_CHUNK", chunk, 1);
            SeqFile f(argv[a]);
            if (!f.can_stream()) { printf("%s cannot stream\n", argv[a]); continue; }
            f.stream_begin(4);
            size_t q = 0, got = 0;
            for (;;) { got = f.stream_wait(q + 7); const bool done = f.stream_done(); q = got; if (done) break; }
            f.stream_end();
            const std::string have = dump(f);
            if (have != want) { ++bad; printf("MISMATCH %s chunk %s: %zu vs %zu records\n", argv[a], chunk, f.records(), e.records()); }
        }
        printf("%s: %zu records ok\n", argv[a], e.records());
    }
    return bad;
}
