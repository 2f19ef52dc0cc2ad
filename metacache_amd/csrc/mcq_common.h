// mcq_common.h -- pieces shared by the modes of mcq (query: mcq_main.cpp, build: mcq_build.h): taxonomy records and rank names
// (taxonomy.hpp:68-255), directory expansion (filesys_utility.cpp:34-75), sequence files (sequence_io.cpp:160-228), sequence id
// extraction from headers and file names (sequence_io.cpp:470-673).  Plain host C++14 above the C ABI.
#ifndef MCQ_COMMON_H_
#define MCQ_COMMON_H_
#include "metacache_amd.h"

#include <dirent.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <regex>
#include <thread>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <cmath>
#include <iomanip>
#include <limits>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace mcq {

constexpr int kNumRanks = MC_NUM_RANKS;      // 21, index 21 = none
const char* const kRankNames[] = {"sequence", "form", "variety", "subspecies", "species", "subgenus", "genus", "subtribe", "tribe",
                                  "subfamily", "family", "suborder", "order", "subclass", "class", "subphylum", "phylum",
                                  "subkingdom", "kingdom", "domain", "root", "none"};

inline int rank_from_name(std::string n)                    // taxonomy.hpp:174-214
{
    std::transform(n.begin(), n.end(), n.begin(), ::tolower);
    for (int i = 0; i <= kNumRanks; ++i) if (n == kRankNames[i]) return i;
    if (n == "genome") return 0;
    return -1;
}

// taxonomy::rank_from_name (taxonomy.hpp:182-221) for the rank column of nodes.dmp: NCBI's names folded onto the 21 ranks; unknown = none
inline int rank_from_dump_name(std::string n)
{
    std::transform(n.begin(), n.end(), n.begin(), ::tolower);
    static const std::pair<const char*, int> table[] = {
        {"sequence", 0}, {"genome", 0}, {"form", 1}, {"forma", 1}, {"variety", 2}, {"varietas", 2}, {"subspecies", 3}, {"species", 4},
        {"species group", 5}, {"species subgroup", 5}, {"subgenus", 5}, {"genus", 6}, {"subtribe", 7}, {"tribe", 8}, {"subfamily", 9},
        {"family", 10}, {"superfamily", 11}, {"parvorder", 11}, {"infraorder", 11}, {"suborder", 11}, {"order", 12}, {"superorder", 13},
        {"infraclass", 13}, {"subclass", 13}, {"class", 14}, {"superclass", 15}, {"subphylum", 15}, {"phylum", 16}, {"division", 16},
        {"superphylum", 17}, {"subkingdom", 17}, {"kingdom", 18}, {"subdomain", 18}, {"superkingdom", 19}, {"domain", 19}, {"root", 20}};
    for (const auto& e : table) if (n == e.first) return e.second;
    return kNumRanks;
}

struct Taxon { int64_t id = 0, parent = 0; int rank = kNumRanks; std::string name; uint64_t windows = 0; };
using Lineage = std::array<uint32_t, kNumRanks>;     // taxon index + 1, 0 = none

struct Taxonomy {
    std::vector<Taxon> taxa;
    std::unordered_map<int64_t, uint32_t> byId;
    const uint32_t* targetLineages = nullptr;        // [targets * 21]
    uint64_t numTargets = 0;

    std::map<std::string, uint32_t> targetByName;    // name2tax_ (taxonomy.hpp:1108-1127): sequence-level taxa by name
    std::vector<char> coveredCache;                 // covers(): taxa on the full lineage of any target

    const Taxon* taxon(uint32_t idxPlus1) const { return idxPlus1 ? &taxa[idxPlus1 - 1] : nullptr; }
    uint32_t with_name(const std::string& n) const { if (n.empty()) return 0; auto i = targetByName.find(n); return i == targetByName.end() ? 0 : i->second; }
    uint32_t with_similar_name(const std::string& n) const
    {
        if (n.empty()) return 0;
        auto i = targetByName.upper_bound(n);
        if (i == targetByName.end() || i->first.compare(0, n.size(), n) != 0) return 0;
        return i->second;
    }
    uint32_t with_id(int64_t id) const { auto i = byId.find(id); return i == byId.end() ? 0 : i->second + 1; }
    // cached_next_ranked_ancestor (taxonomy.hpp:1245-1256)
    uint32_t next_ranked_ancestor(uint32_t t) const
    {
        if (!t) return 0;
        if (taxon(t)->rank != kNumRanks) return t;
        for (uint32_t a : ranks_of(t)) if (a) return a;
        return 0;
    }
    bool covers(uint32_t t) const { return t && coveredCache[t]; }   // taxonomy.hpp:1355-1366
    void build_covered()
    {
        {
            coveredCache.assign(taxa.size() + 1, 0);
            for (size_t i = 0; i < taxa.size(); ++i) {
                if (taxa[i].rank != 0 || taxa[i].id >= 0) continue;          // targets only
                coveredCache[i + 1] = 1;
                int64_t id = taxa[i].parent;
                for (int guard = 0; id != 0 && guard < 1000; ++guard) {
                    auto it = byId.find(id);
                    if (it == byId.end()) break;
                    coveredCache[it->second + 1] = 1;
                    if (taxa[it->second].parent == id) break;
                    id = taxa[it->second].parent;
                }
            }
        }
    }
    Lineage target_ranks(uint32_t tgt) const
    {
        Lineage l{};
        if (tgt < numTargets) std::copy(targetLineages + (size_t)tgt * kNumRanks, targetLineages + (size_t)(tgt + 1) * kNumRanks, l.begin());
        return l;
    }
    // taxonomy::make_ranks (taxonomy.hpp:576-597)
    Lineage ranks_of(uint32_t idxPlus1) const
    {
        Lineage l{};
        const Taxon* t = taxon(idxPlus1);
        if (!t) return l;
        if (t->rank < kNumRanks) l[t->rank] = idxPlus1;
        int64_t id = t->parent;
        while (id != 0) {
            auto it = byId.find(id);
            if (it == byId.end()) break;
            const Taxon& p = taxa[it->second];
            if (p.rank < kNumRanks) l[p.rank] = it->second + 1;
            if (p.parent == id) break;
            id = p.parent;
        }
        return l;
    }
};


// files_in_directory (filesys_utility.cpp:34-75): entries in readdir order, directories expanded at most 'recurse' levels deep
inline std::vector<std::string> files_in_directory(std::string dirName, int recurse = 10)
{
    while (!dirName.empty() && (dirName.back() == '/' || dirName.back() == '\\')) { dirName.pop_back(); break; }
    std::vector<std::string> files;
    if (DIR* dir = opendir(dirName.c_str())) {
        while (dirent* e = readdir(dir)) {
            const std::string nm = e->d_name;
            if (nm == "." || nm == "..") continue;
            const std::string path = dirName + "/" + nm;
            std::vector<std::string> sub;
            if (recurse > 0) sub = files_in_directory(path, recurse - 1);
            if (sub.empty()) files.push_back(path); else files.insert(files.end(), sub.begin(), sub.end());
        }
        closedir(dir);
    }
    return files;
}


// ---- sequence files (FASTA / 4-line FASTQ; plain = memory-mapped, gzip = inflated into memory) ---------------------------------------------
struct View { const char* p = nullptr; size_t n = 0; bool empty() const { return n == 0; } };

class SeqFile {
public:
    explicit SeqFile(const std::string& fn)
    {
        fd_ = ::open(fn.c_str(), O_RDONLY);
        struct stat st;
        if (fd_ < 0 || fstat(fd_, &st) != 0) throw std::runtime_error("file '" + fn + "' could not be opened");
        unsigned char magic[2] = {0, 0};
        const bool gz = st.st_size >= 2 && pread(fd_, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
        if (gz) {
            // compressed input (the reference reads it through zlib as well): inflated into memory, then handled like a mapped file
            gzFile g = gzdopen(dup(fd_), "rb");
            if (!g) throw std::runtime_error("file '" + fn + "' could not be opened");
            gzbuffer(g, 1u << 20);
            size_t cap = std::max<size_t>((size_t)st.st_size * 4, 1u << 20);
            own_.resize(cap);
            for (;;) {
                if (size_ == own_.size()) own_.resize(own_.size() * 2);
                const int got = gzread(g, own_.data() + size_, (unsigned)std::min<size_t>(own_.size() - size_, 1u << 30));
                if (got < 0) { gzclose(g); throw std::runtime_error("file '" + fn + "' could not be decompressed"); }
                if (got == 0) break;
                size_ += (size_t)got;
            }
            gzclose(g);
            data_ = own_.data();
            return;
        }
        size_ = (size_t)st.st_size;
        if (size_) {
            void* m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
            if (m == MAP_FAILED) throw std::runtime_error("file '" + fn + "' could not be mapped");
            data_ = (const char*)m;
            mapped_ = true;
            madvise(m, size_, MADV_SEQUENTIAL);
        }
    }
    ~SeqFile() { for (auto& t : streamPool_) if (t.joinable()) t.join(); if (mapped_) munmap((void*)data_, size_); if (fd_ >= 0) ::close(fd_); }
    SeqFile(const SeqFile&) = delete;

    // record starts = lines beginning with '>' (FASTA) or '@' header lines of 4-line FASTQ records; found by all threads
    void index(unsigned threads)
    {
        const size_t first = skip_stray(0);                                       // sequence_io.cpp:168-173: skip to the first '>' / '@' line
        if (first >= size_) return;
        fastq_ = data_[first] == '@';
        threads = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, size_ / (1u << 22) + 1));
        std::vector<std::vector<uint64_t>> found(threads);
        std::vector<std::thread> pool;
        const size_t span = (size_ - first + threads - 1) / threads;
        for (unsigned t = 0; t < threads; ++t)
            pool.emplace_back([&, t] { scan(first + t * span, std::min(size_, first + (t + 1) * span), found[t], irregular_); });
        for (auto& th : pool) th.join();
        if (irregular_.load()) { scan_exact(first); return; }
        for (auto& v : found) starts_.insert(starts_.end(), v.begin(), v.end());
    }
    size_t records() const { return streaming_ ? streamCount_.load() + recs_.size() : (exact_ ? recs_.size() : starts_.size()); }

    // sequence_reader::read_next: header without the marker, sequence lines joined ('scratch' only for multi-line records)
    void record(size_t i, View& header, View& seq, std::string& scratch) const
    {
        size_t b, e;
        bool exactRec = exact_;
        if (streaming_) {
            const size_t sc = streamCount_.load(std::memory_order_acquire);
            if (i >= sc) { b = recs_[i - sc].start; e = recs_[i - sc].seqEnd; exactRec = true; }
            else { b = stream_start(i); e = i + 1 < sc ? stream_start(i + 1) : streamEnd_.load(std::memory_order_acquire); exactRec = false; }
        }
        else if (exact_) { b = recs_[i].start; e = recs_[i].seqEnd; }
        else { b = starts_[i]; e = i + 1 < starts_.size() ? starts_[i + 1] : size_; }
        size_t eol = line_end(b, e);
        header = trimmed(b + 1, eol);
        size_t p = std::min(eol + 1, e);
        if (fastq_ && !exactRec) { seq = trimmed(p, line_end(p, e)); return; }
        seq = View{};
        bool multi = false;
        while (p < e) {
            eol = line_end(p, e);
            const View l = trimmed(p, eol);
            if (l.n) {
                if (seq.n == 0 && !multi) seq = l;
                else { if (!multi) { scratch.assign(seq.p, seq.n); multi = true; } scratch.append(l.p, l.n); }
            }
            p = eol + 1;
        }
        if (multi) seq = View{scratch.data(), scratch.size()};
    }

    // ---- streaming index (plain files): chunks of the file are scanned by a few threads in file order and published one after the
    // other, so that batches of the first chunks are on the device while the last ones are still being scanned (the scan is the first
    // touch of the mapping: 40 ms for 1.7 GB).  A chunk with irregular input stops the publication; the exact sequential scan takes
    // over from the last record start that is certain.
    static size_t env_size(const char* name, size_t dflt) { const char* e = std::getenv(name); return e ? (size_t)std::strtoull(e, nullptr, 10) : dflt; }
    bool can_stream() const { return mapped_ && size_ > env_size("MCQ_STREAM_MIN", 8u << 20); }   // (tests lower both to reach this path with toy files)
    void stream_begin(unsigned threads)
    {
        streaming_ = true;
        const size_t first = skip_stray(0);
        streamEnd_ = size_;
        if (first >= size_) { streamDone_ = true; return; }
        fastq_ = data_[first] == '@';
        const size_t kChunk = std::max<size_t>(64, env_size("MCQ_STREAM_CHUNK", 8u << 20));
        const size_t nch = (size_ - first + kChunk - 1) / kChunk;
        chunkStarts_.assign(nch, std::vector<uint64_t>());
        chunkFirst_.assign(nch + 1, 0);
        chunkState_ = std::vector<std::atomic<int>>(nch);
        for (auto& c : chunkState_) c.store(0);
        streamFirstByte_ = first;
        threads = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, nch));
        for (unsigned t = 0; t < threads; ++t)
            streamPool_.emplace_back([this, nch, kChunk] {
                for (size_t c; (c = nextChunk_++) < nch;) {
                    std::atomic<bool> irr{false};
                    const size_t lo = streamFirstByte_ + c * kChunk, hi = std::min(size_, lo + kChunk);
                    scan(lo, hi, chunkStarts_[c], irr);
                    chunkState_[c].store(irr.load() ? 2 : 1, std::memory_order_release);
                    { std::lock_guard<std::mutex> l(streamMtx_); }
                    streamCv_.notify_all();
                }
            });
    }
    // blocks until `want` records can be READ (start and end known) or everything is indexed; returns the number that can be read
    size_t stream_wait(size_t want)
    {
        std::unique_lock<std::mutex> l(streamMtx_);
        for (;;) {
            // publish the chunks that are through, in order
            while (!streamDone_ && published_.load() < chunkStarts_.size()) {
                const int st = chunkState_[published_.load()].load(std::memory_order_acquire);
                if (st == 0) break;
                if (st == 2) { finish_exact_locked(l); break; }
                const size_t c = published_.load();
                chunkFirst_[c + 1] = chunkFirst_[c] + chunkStarts_[c].size();
                published_.store(c + 1, std::memory_order_release);
                streamCount_.store(chunkFirst_[c + 1]);
                if (c + 1 == chunkStarts_.size()) streamDone_ = true;
            }
            const size_t sc = streamCount_.load();
            const size_t readable = streamDone_ ? records() : (sc ? sc - 1 : 0);   // the last one's end is the next start
            if (streamDone_ || readable >= want) return readable;
            streamCv_.wait(l);
        }
    }
    bool stream_done() { std::lock_guard<std::mutex> l(streamMtx_); return streamDone_; }
    void stream_end() { for (auto& t : streamPool_) t.join(); streamPool_.clear(); }

private:
    // irregular chunk: what was published stays (its last record is dropped: its end is not certain), the rest of the file goes
    // through the exact scan, from that record's start
    void finish_exact_locked(std::unique_lock<std::mutex>& l)
    {
        nextChunk_.store(chunkStarts_.size());                                   // no further chunks are taken
        l.unlock();
        for (auto& t : streamPool_) t.join();
        streamPool_.clear();
        l.lock();
        size_t from = streamFirstByte_;
        // the end of the last record that stays published first, then the smaller count: a worker that reads the new count must
        // see the new end (record(): e = streamEnd_ for the last streamed record)
        const size_t sc = streamCount_.load();
        if (sc > 0) from = stream_start(sc - 1);
        streamEnd_.store(from, std::memory_order_release);
        if (sc > 0) streamCount_.store(sc - 1, std::memory_order_release);
        scan_exact(from, /*keepMode=*/true);                                     // exact_ is read by the workers: not touched while streaming
        streamDone_ = true;
    }
    size_t stream_start(size_t i) const
    {
        size_t lo = 0, hi = published_.load(std::memory_order_acquire);                                          // chunk c: chunkFirst_[c] <= i < chunkFirst_[c + 1]
        while (hi - lo > 1) { const size_t mid = (lo + hi) / 2; if (chunkFirst_[mid] <= i) lo = mid; else hi = mid; }
        return (size_t)chunkStarts_[lo][i - chunkFirst_[lo]];
    }
    bool streaming_ = false, streamDone_ = false;
    std::atomic<size_t> streamCount_{0}, published_{0};
    std::atomic<size_t> streamEnd_{0};
    size_t streamFirstByte_ = 0;
    std::vector<std::vector<uint64_t>> chunkStarts_;
    std::vector<uint64_t> chunkFirst_;
    std::vector<std::atomic<int>> chunkState_;
    std::atomic<size_t> nextChunk_{0};
    std::vector<std::thread> streamPool_;
    std::mutex streamMtx_;
    std::condition_variable streamCv_;
    size_t next_line(size_t p) const { const void* nl = memchr(data_ + p, '\n', size_ - p); return nl ? (size_t)((const char*)nl - data_) + 1 : size_; }
    size_t line_end(size_t p, size_t e) const { if (p >= e) return e; const void* nl = memchr(data_ + p, '\n', e - p); return nl ? (size_t)((const char*)nl - data_) : e; }
    View trimmed(size_t b, size_t e) const { while (e > b && (data_[e - 1] == '\r' || data_[e - 1] == '\n')) --e; return View{data_ + b, e > b ? e - b : 0}; }
    bool fastq_header_at(size_t p) const
    {
        if (p >= size_ || data_[p] != '@') return false;
        const size_t l2 = next_line(next_line(p));
        return l2 < size_ && data_[l2] == '+';
    }
    // The fast scans below assume well-formed input: FASTA without lines that begin with '+', FASTQ as strict 4-line records.
    // Anything else (sequences over several lines in FASTQ, stray lines, records of both kinds in one file) is what the reference's
    // reader handles with ONE sequential state machine (sequence_reader::read_next, sequence_io.cpp:160-236): a scan that meets such
    // input raises irregular_, and index() repeats the job with scan_exact(), which restates that machine line by line.
    void scan(size_t lo, size_t hi, std::vector<uint64_t>& out, std::atomic<bool>& irregular_)
    {
        if (!fastq_) {
            for (size_t p = lo; p < hi;) {
                const void* g = memchr(data_ + p, '>', hi - p);
                if (!g) break;
                const size_t q = (size_t)((const char*)g - data_);
                if (q == 0 || data_[q - 1] == '\n') out.push_back(q);
                p = q + 1;
            }
            for (size_t p = lo; p < hi;) {                                         // a line that begins with '+' ends a record there
                const void* g = memchr(data_ + p, '+', hi - p);
                if (!g) break;
                const size_t q = (size_t)((const char*)g - data_);
                if (q == 0 || data_[q - 1] == '\n') { irregular_.store(true); return; }
                p = q + 1;
            }
            return;
        }
        size_t p = lo;
        if (p > 0 && data_[p - 1] != '\n') p = next_line(p);
        while (p < hi && !fastq_header_at(p)) p = next_line(p);                   // an '@' line whose line+2 starts with '+' is a header
        while (p < hi) {
            out.push_back(p);
            const size_t l1 = next_line(p), l2 = next_line(l1), l3 = next_line(l2);
            // strict record: header, ONE non-empty sequence line that is not a marker line, '+' line, quality line
            if (l1 >= size_ || data_[l1] == '+' || data_[l1] == '>' || data_[l1] == '\n' || data_[l1] == '\r' || l2 >= size_ || data_[l2] != '+') {
                if (l1 < size_) { irregular_.store(true); return; }
            }
            p = next_line(l3);
            if (p < size_ && data_[p] != '@') { irregular_.store(true); return; } // stray lines, FASTA records: the exact scan decides
        }
    }
    // sequence_reader::read_next restated: a record begins at the next line that starts with '>' or '@'; its sequence is every
    // non-empty line up to a line that starts with '>' or '+'; after a '+' line exactly one more line (the qualities) is dropped.
    void scan_exact(size_t first, bool keepMode = false)
    {
        if (!keepMode) exact_ = true;
        starts_.clear();
        size_t p = first;
        while (p < size_) {
            p = skip_stray(p);
            if (p >= size_) break;
            Rec r; r.start = p;
            p = next_line(p);
            while (p < size_ && data_[p] != '>' && data_[p] != '+') p = next_line(p);   // empty lines are skipped when the record is read
            r.seqEnd = p;
            recs_.push_back(r);
            if (p < size_ && data_[p] == '+') p = next_line(next_line(p));          // '+' line and ONE quality line
        }
    }
    // lines that do not begin a record are dropped (read_next :168-173).  Mirrored detail: the reader takes a line's first character
    // and then skips "the rest of the line" -- for an EMPTY line the character it took was the newline itself, so the skip swallows
    // the FOLLOWING line, whatever it holds (a record header after a blank stray line is lost).
    size_t skip_stray(size_t p) const
    {
        while (p < size_ && data_[p] != '>' && data_[p] != '@') p = data_[p] == '\n' ? next_line(p + 1) : next_line(p);
        return p;
    }
    struct Rec { size_t start, seqEnd; };
    bool exact_ = false;
    std::atomic<bool> irregular_{false};
    std::vector<Rec> recs_;
    int fd_ = -1;
    const char* data_ = nullptr;
    size_t size_ = 0;
    bool fastq_ = false, mapped_ = false;
    std::vector<char> own_;
    std::vector<uint64_t> starts_;
};

// ---- sequence ids and taxon ids in headers / file names (sequence_io.cpp:470-673) ----
inline const std::regex& accession_regex()
{
    static const std::regex re("(^|[^[:alnum:]])(([A-Z][_A-Z]{1,9}[0-9]{5,})(\\.[0-9]+)?)", std::regex::optimize);
    return re;
}

inline std::string leading_word(const std::string& t)
{
    auto fst = std::find_if(t.begin(), t.end(), [](char c) { return !std::isspace((unsigned char)c); });
    if (fst == t.end()) return t;
    auto lst = std::find_if(fst + 1, t.end(), [](char c) { return std::isspace((unsigned char)c); });
    return std::string(fst, lst);
}

inline std::string filename_without_extension(const std::string& t)
{
    if (t.empty()) return t;
    auto fst = std::find(t.rbegin(), t.rend(), '/').base();
    auto ext = std::find(fst, t.end(), '.');
    return std::string(fst, ext);
}

inline int64_t taxon_id_in_header(const std::string& t)
{
    auto i = t.find("taxid");
    if (i == std::string::npos) return 0;
    i += 6;                                                   // "taxid" + one separator character
    if (i > t.size()) return 0;
    auto j = t.find('|', i);
    if (j == std::string::npos) { j = t.find(' ', i); if (j == std::string::npos) j = t.size(); }
    try { return (int64_t)std::stoull(t.substr(i, j - i)); } catch (std::exception&) { return 0; }
}

}  // namespace mcq
#endif
