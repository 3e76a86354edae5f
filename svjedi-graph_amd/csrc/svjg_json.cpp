// libsvjg_host.so — host-side writer of <prefix>_informative_aln.json (no GPU code in this file).
//
// Reproduces, byte for byte, what the reference writes with
//     json.dumps(dict_of_informative_aln, sort_keys=True, indent=4)        (filter-alignments.py:174-175)
// from the device's hit records: for every SV key (sorted by code point = UTF-8 byte order) the two lists of
// alignment texts in file order, where a text is the line as Python's text mode delivers it (terminator
// translated to "\n", absent at EOF) cut before the first "cg:Z:" (filter-alignments.py:166), escaped like
// json's ensure_ascii=True.  Multi-threaded by key ranges; every thread writes its own ranges at their final offsets.
#include "../../include/svjg.h"
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Job {
    const uint8_t *gaf; uint64_t n;
    const svjg_hitrec *recs;                 // grouped by slot, each group sorted by line_start
    const uint64_t *slot_begin;              // n_slots + 1
    const char *const *sv_ids;
    const uint32_t *order; uint64_t n_keys;  // slots with >= 1 record, sorted by key
};

inline char *hex4(char *w, uint32_t v) {
    static const char *H = "0123456789abcdef";
    w[0] = '\\'; w[1] = 'u'; w[2] = H[(v >> 12) & 15]; w[3] = H[(v >> 8) & 15]; w[4] = H[(v >> 4) & 15]; w[5] = H[v & 15];
    return w + 6;
}

// what json.dumps(ensure_ascii=True) does with a byte < 0x80: 0 = stands for itself, 1 = \uXXXX, else the letter after the backslash
struct EscTable {
    uint8_t t[256];
    EscTable() {
        for (int c = 0; c < 256; ++c) t[c] = (c < 0x20 || c == 0x7F) ? 1 : 0;     // json escapes everything outside ' '..'~'
        t[(int)'"'] = '"'; t[(int)'\\'] = '\\'; t[(int)'\n'] = 'n'; t[(int)'\r'] = 'r'; t[(int)'\t'] = 't'; t[(int)'\b'] = 'b'; t[(int)'\f'] = 'f';
        for (int c = 0x80; c < 256; ++c) t[c] = 2;                               // start of a multi-byte sequence (or garbage)
    }
};
const EscTable ESC;

// JSON string body with ensure_ascii=True written to w (room for the worst case, six characters per byte, is the
// caller's business); returns the end of the output, nullptr on malformed UTF-8.
char *escape(char *w, const uint8_t *s, size_t n) {
    for (size_t i = 0; i < n;) {
        const uint8_t c = s[i];
        const uint8_t k = ESC.t[c];
        if (k == 0) { *w++ = (char)c; ++i; continue; }
        if (c < 0x80) {
            if (k == 1) w = hex4(w, c); else { w[0] = '\\'; w[1] = (char)k; w += 2; }
            ++i;
            continue;
        }
        uint32_t cp; size_t len;
        if ((c & 0xE0) == 0xC0) { cp = c & 0x1F; len = 2; }
        else if ((c & 0xF0) == 0xE0) { cp = c & 0x0F; len = 3; }
        else if ((c & 0xF8) == 0xF0) { cp = c & 0x07; len = 4; }
        else return nullptr;
        if (i + len > n) return nullptr;
        for (size_t q = 1; q < len; ++q) { if ((s[i + q] & 0xC0) != 0x80) return nullptr; cp = (cp << 6) | (s[i + q] & 0x3F); }
        if ((len == 2 && cp < 0x80) || (len == 3 && cp < 0x800) || (len == 4 && cp < 0x10000) || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) return nullptr;
        if (cp >= 0x10000) { cp -= 0x10000; w = hex4(w, 0xD800 + (cp >> 10)); w = hex4(w, 0xDC00 + (cp & 0x3FF)); }
        else w = hex4(w, cp);
        i += len;
    }
    return w;
}

// per-thread scratch for one escaped string: grows, never shrinks, never refilled
struct Scratch {
    std::vector<char> buf; size_t len = 0;
    char *room(size_t n) { if (buf.size() < n) buf.resize(n + n / 2 + 64); return buf.data(); }
};

// escaped text of the line starting at `start` -> o
bool line_text(const Job &j, uint64_t start, Scratch &o) {
    const uint8_t *p = j.gaf + start;
    const size_t room = (size_t)(j.n - start);
    const uint8_t *nl = (const uint8_t *)memchr(p, '\n', room);
    size_t len = nl ? (size_t)(nl - p) : room;
    if (const uint8_t *cr = (const uint8_t *)memchr(p, '\r', len)) len = (size_t)(cr - p);      // a lone \r or \r\n ends the line too
    const bool at_end = start + len >= j.n;
    // cut before the first "cg:Z:" (searched in the translated text: the tag cannot span the terminator)
    const uint8_t *tag = len >= 5 ? (const uint8_t *)memmem(p, len, "cg:Z:", 5) : nullptr;
    const size_t take = tag ? (size_t)(tag - p) : len;
    char *w0 = o.room(6 * take + 2), *w = escape(w0, p, take);
    if (!w) return false;
    if (!tag && !at_end) { w[0] = '\\'; w[1] = 'n'; w += 2; }
    o.len = (size_t)(w - w0);
    return true;
}

bool render_key(const Job &j, uint64_t ki, std::string &o, Scratch &tmp) {
    const uint32_t slot = j.order[ki];
    o += ki ? ",\n    \"" : "\n    \"";
    const char *id = j.sv_ids[slot];
    { const size_t n = strlen(id);
      char *w0 = tmp.room(6 * n + 1), *w = escape(w0, (const uint8_t *)id, n);
      if (!w) return false;
      o.append(w0, (size_t)(w - w0)); }
    o += "\": [";
    for (int allele = 0; allele < 2; ++allele) {
        uint64_t cnt = 0;
        for (uint64_t r = j.slot_begin[slot]; r < j.slot_begin[slot + 1]; ++r) {
            const uint32_t rep = allele ? j.recs[r].n_alt : j.recs[r].n_ref;
            if (!rep) continue;
            if (!line_text(j, j.recs[r].line_start, tmp)) return false;
            for (uint32_t k = 0; k < rep; ++k) {
                o += cnt ? ",\n            \"" : "\n        [\n            \"";
                o.append(tmp.buf.data(), tmp.len); o += '"';
                ++cnt;
            }
        }
        if (cnt) o += "\n        ]"; else o += "\n        []";
        if (!allele) o += ",";
    }
    o += "\n    ]";
    return true;
}

}  // namespace

// recs: n_recs hit records (any order; line_start relative to `gaf`); sv_ids[slot] = key string of each count slot.
// Returns 0, SVJG_E_ARG, SVJG_E_NOMEM (cannot open / write) or SVJG_E_INPUT (text is not valid UTF-8).
extern "C" int svjg_write_informative_json(const char *path, const char *gaf, uint64_t n_bytes, const svjg_hitrec *recs_in,
                                           uint64_t n_recs, const char *const *sv_ids, uint32_t n_slots, int n_threads)
{
    if (!path || (n_bytes && !gaf) || (n_recs && !recs_in) || (n_slots && !sv_ids)) return SVJG_E_ARG;
    const bool verbose = getenv("SVJG_VERBOSE") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    // group by slot (counting sort, the records cut into one range per thread), then file order inside each group
    int T = n_threads > 0 ? n_threads : (int)std::min(64u, std::thread::hardware_concurrency());
    if (T < 1) T = 1;
    const int G = (int)std::min<uint64_t>((uint64_t)std::min(T, 16), n_recs / 65536 + 1);
    std::vector<uint64_t> begin((size_t)n_slots + 1, 0);
    std::vector<svjg_hitrec> recs(n_recs);
    {
        std::vector<std::vector<uint64_t>> hist((size_t)G, std::vector<uint64_t>((size_t)n_slots + 1, 0));
        std::atomic<int> bad_slot{0};
        auto range = [&](int g, uint64_t &lo, uint64_t &hi) { lo = n_recs * (uint64_t)g / (uint64_t)G; hi = n_recs * (uint64_t)(g + 1) / (uint64_t)G; };
        auto each = [&](auto &&f) {
            std::vector<std::thread> th;
            for (int g = 1; g < G; ++g) th.emplace_back(f, g);
            f(0);
            for (auto &x : th) x.join();
        };
        each([&](int g) {
            uint64_t lo, hi; range(g, lo, hi);
            uint64_t *h = hist[(size_t)g].data();
            for (uint64_t i = lo; i < hi; ++i) { if (recs_in[i].slot >= n_slots) { bad_slot.store(1); return; } h[recs_in[i].slot]++; }
        });
        if (bad_slot.load()) return SVJG_E_ARG;
        uint64_t acc = 0;                                      // hist[g][s] becomes where range g puts its first record of slot s
        for (uint32_t sl = 0; sl < n_slots; ++sl) {
            begin[sl] = acc;
            for (int g = 0; g < G; ++g) { const uint64_t c = hist[(size_t)g][sl]; hist[(size_t)g][sl] = acc; acc += c; }
        }
        begin[n_slots] = acc;
        each([&](int g) {
            uint64_t lo, hi; range(g, lo, hi);
            uint64_t *cur = hist[(size_t)g].data();
            for (uint64_t i = lo; i < hi; ++i) recs[cur[recs_in[i].slot]++] = recs_in[i];
        });
    }
    std::vector<uint32_t> order;
    for (uint32_t s = 0; s < n_slots; ++s) if (begin[s + 1] > begin[s]) order.push_back(s);
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return strcmp(sv_ids[a], sv_ids[b]) < 0; });

    Job job{(const uint8_t *)gaf, n_bytes, recs.data(), begin.data(), sv_ids, order.data(), order.size()};
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) return SVJG_E_NOMEM;
    auto put = [&](const char *p, size_t n, uint64_t at) {                 // pwrite all of it
        while (n) {
            ssize_t w = pwrite(fd, p, n, (off_t)at);
            if (w <= 0) return false;
            p += w; n -= (size_t)w; at += (uint64_t)w;
        }
        return true;
    };
    if (order.empty()) { bool ok = put("{}", 2, 0); return (close(fd) || !ok) ? SVJG_E_NOMEM : 0; }

    // tasks = runs of keys with about REC_PER_TASK hit records (a few MB of text each).  Workers render tasks into memory
    // in parallel; ONE thread (the caller) writes the rendered tasks to the file in order.  Writes to one file are
    // serialised by the kernel (inode lock): a single writer reaches 9 GB/s on tmpfs and 14 GB/s on a disk file system's
    // page cache, 64 concurrent pwrite()rs 4 GB/s (tools/ubench/writetest.cpp), so the workers only render.
    const uint64_t REC_PER_TASK = 16384;
    std::vector<uint64_t> task_lo{0};
    { uint64_t acc = 0;
      for (uint64_t ki = 0; ki < order.size(); ++ki) {
          acc += begin[order[ki] + 1] - begin[order[ki]];
          if (acc >= REC_PER_TASK && ki + 1 < order.size()) { task_lo.push_back(ki + 1); acc = 0; }
      }
      task_lo.push_back(order.size()); }
    const uint64_t n_tasks = task_lo.size() - 1;
    if (T > 1) --T;                                            // the caller's thread writes
    if (T < 1) T = 1;
    if ((uint64_t)T > n_tasks) T = (int)n_tasks;
    const uint64_t window = 2 * (uint64_t)T + 8;               // rendered-but-unwritten tasks are bounded: memory
    std::vector<std::string> text(n_tasks);
    std::vector<char> done(n_tasks, 0);
    std::atomic<uint64_t> next{0};
    std::atomic<int> bad{0};                                   // 1 = text is not valid UTF-8, 2 = write error
    std::mutex mu; std::condition_variable cv;
    uint64_t written = 0;                                      // tasks 0 .. written-1 are in the file
    std::vector<std::string> pool;                             // written-out buffers go back to the workers (their capacity is kept)
    auto worker = [&]() {
        Scratch tmp;
        std::string s;
        for (;;) {
            const uint64_t t = next.fetch_add(1);
            if (t >= n_tasks) return;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return t < written + window || bad.load(); });
              if (s.capacity() < 4096 && !pool.empty()) { s.swap(pool.back()); pool.pop_back(); } }
            if (bad.load()) return;
            s.clear();
            for (uint64_t ki = task_lo[t]; ki < task_lo[t + 1]; ++ki) {
                const uint32_t slot = order[ki];
                // every line of every group sorted by file order
                std::sort(recs.begin() + begin[slot], recs.begin() + begin[slot + 1],
                          [](const svjg_hitrec &a, const svjg_hitrec &b) { return a.line_start < b.line_start; });
                if (!render_key(job, ki, s, tmp)) { bad.store(1); break; }
            }
            { std::unique_lock<std::mutex> lk(mu);
              text[t].swap(s); done[t] = 1;
              cv.notify_all(); }
            if (bad.load()) return;
        }
    };
    int rc = 0;
    if (!put("{", 1, 0)) rc = SVJG_E_NOMEM;
    uint64_t at = 1;
    double waited = 0;
    const auto t_mid = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    if (!rc) for (int i = 0; i < T; ++i) th.emplace_back(worker);
    for (uint64_t t = 0; !rc && t < n_tasks; ++t) {
        std::string s;
        const auto w0 = std::chrono::steady_clock::now();
        { std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return done[t] || bad.load(); });
          if (!done[t]) break;
          s.swap(text[t]); }
        waited += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
        if (!put(s.data(), s.size(), at)) { bad.store(2); cv.notify_all(); break; }
        at += s.size();
        s.clear();
        { std::unique_lock<std::mutex> lk(mu); written = t + 1; pool.emplace_back(std::move(s)); cv.notify_all(); }
    }
    for (auto &x : th) x.join();
    if (!rc && bad.load()) rc = bad.load() == 1 ? SVJG_E_INPUT : SVJG_E_NOMEM;
    if (!rc && !put("\n}", 2, at)) rc = SVJG_E_NOMEM;
    if (close(fd) && !rc) rc = SVJG_E_NOMEM;
    if (verbose) {
        const auto t_end = std::chrono::steady_clock::now();
        fprintf(stderr, "[svjg] json writer: group + sort %.2f s, render + write %.2f s (%d rendering threads, one writer that waited %.2f s for them, %llu tasks)\n",
                std::chrono::duration<double>(t_mid - t_start).count(), std::chrono::duration<double>(t_end - t_mid).count(), T, waited, (unsigned long long)n_tasks);
    }
    return rc;
}

// ---------------------------------------------------------------------------------------------------------------
// Reader side: predict-genotype.py only needs len(dict[key][0]) and len(dict[key][1]) (predict-genotype.py:219-226).
// svjg_count_informative_json scans the JSON text once (any valid JSON of that shape, not only json.dumps' layout)
// and returns the keys (unescaped, UTF-8, NUL separated) and the two list lengths per key.
//
// r06: (1) strings are skipped eight bytes per step and read the way json.load reads them (strict: a raw control character or an escape
// other than \" \\ \/ \b \f \n \r \t \uXXXX is an error — the reference dies with JSONDecodeError there, predict-genotype.py:67-68);
// (2) a file of 64 MB and more is parsed by several threads, exactly: the text is a sequence of PAIRS  "key": [[...], [...]]  separated
// by commas.  Thread i starts at a guessed pair start — the first line at or behind its share of the file that begins like json.dumps'
// key lines, four blanks and a quote — and parses pairs with the one parser below until it arrives at thread i + 1's start.  A guess is
// never trusted: the result is taken only if the CHAIN holds — thread 0 starts where the object starts, and every thread ends, behind a
// comma, exactly where the next one started, which means the sequential parser would have stood there expecting a pair too — else
// (another layout, a guess inside a pair) one thread parses the whole file as before.  11.6 GB: 18 s -> ~1.5 s; the 117 GB of configs[3] go
// through the contract path of predict-genotype.py (no counts hand-off) in seconds instead of minutes.
// ---------------------------------------------------------------------------------------------------------------

namespace {

struct Scan {
    const uint8_t *p, *e;
    bool fail = false;
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p; }
    bool eat(char c) { ws(); if (p < e && *p == (uint8_t)c) { ++p; return true; } return false; }
    static bool hex4(const uint8_t *q, uint32_t &v) {
        v = 0;
        for (int i = 0; i < 4; ++i) {
            const uint8_t h = q[i]; v <<= 4;
            if (h >= '0' && h <= '9') v |= h - '0'; else if (h >= 'a' && h <= 'f') v |= h - 'a' + 10; else if (h >= 'A' && h <= 'F') v |= h - 'A' + 10; else return false;
        }
        return true;
    }
    // the next byte of a string body that is not plain ASCII text: '"', '\\', a control character (< 0x20) or a byte >= 0x80; eight bytes per step
    void plain_run() {
        while (p + 8 <= e) {
            uint64_t w; memcpy(&w, p, 8);
            const uint64_t q = w ^ 0x2222222222222222ull, b = w ^ 0x5C5C5C5C5C5C5C5Cull, c = w & 0xE0E0E0E0E0E0E0E0ull;
            const uint64_t m = ((((q - 0x0101010101010101ull) & ~q) | ((b - 0x0101010101010101ull) & ~b) | ((c - 0x0101010101010101ull) & ~c)) | w) & 0x8080808080808080ull;
            if (m) { p += __builtin_ctzll(m) >> 3; return; }     // (the LOWEST flag of each of the three tests is exact — below the first byte >= 0x80 no borrow reaches them —, so the lowest of all is)
            p += 8;
        }
        while (p < e && *p != '"' && *p != '\\' && *p >= 0x20 && *p < 0x80) ++p;
    }
    // one UTF-8 sequence at p (its first byte is >= 0x80), as Python's decoder takes it — the reference reads the file in text mode and dies with
    // UnicodeDecodeError on anything else: no overlong forms, no surrogates, nothing beyond U+10FFFF
    bool utf8_seq() {
        const uint8_t c = *p;
        const size_t left = (size_t)(e - p);
        if (c >= 0xC2 && c <= 0xDF) { if (left < 2 || (p[1] & 0xC0) != 0x80) return false; p += 2; return true; }
        if ((c & 0xF0) == 0xE0) {
            if (left < 3 || (p[1] & 0xC0) != 0x80 || (p[2] & 0xC0) != 0x80) return false;
            if ((c == 0xE0 && p[1] < 0xA0) || (c == 0xED && p[1] >= 0xA0)) return false;
            p += 3; return true;
        }
        if (c >= 0xF0 && c <= 0xF4) {
            if (left < 4 || (p[1] & 0xC0) != 0x80 || (p[2] & 0xC0) != 0x80 || (p[3] & 0xC0) != 0x80) return false;
            if ((c == 0xF0 && p[1] < 0x90) || (c == 0xF4 && p[1] >= 0x90)) return false;
            p += 4; return true;
        }
        return false;
    }
    // string body -> out (unescaped UTF-8) or skipped when out == nullptr
    bool str(std::string *out) {
        ws();
        if (p >= e || *p != '"') return false;
        ++p;
        for (;;) {
            const uint8_t *q = p;
            plain_run();
            if (out && p > q) out->append((const char *)q, (size_t)(p - q));
            if (p >= e) return false;
            if (*p == '"') { ++p; return true; }
            if (*p >= 0x80) { const uint8_t *u = p; if (!utf8_seq()) return false; if (out) out->append((const char *)u, (size_t)(p - u)); continue; }
            if (*p != '\\') return false;                        // a raw control character (json.load: "Invalid control character")
            if (p + 1 >= e) return false;
            const uint8_t c = p[1];
            p += 2;
            if (c == 'u') {
                if (p + 4 > e) return false;
                uint32_t cp;
                if (!hex4(p, cp)) return false;
                p += 4;
                if (cp >= 0xD800 && cp <= 0xDBFF && p + 6 <= e && p[0] == '\\' && p[1] == 'u') {
                    uint32_t lo;
                    if (hex4(p + 2, lo) && lo >= 0xDC00 && lo <= 0xDFFF) { cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00); p += 6; }
                }
                if (out) {
                    if (cp < 0x80) *out += (char)cp;
                    else if (cp < 0x800) { *out += (char)(0xC0 | (cp >> 6)); *out += (char)(0x80 | (cp & 0x3F)); }
                    else if (cp < 0x10000) { *out += (char)(0xE0 | (cp >> 12)); *out += (char)(0x80 | ((cp >> 6) & 0x3F)); *out += (char)(0x80 | (cp & 0x3F)); }
                    else { *out += (char)(0xF0 | (cp >> 18)); *out += (char)(0x80 | ((cp >> 12) & 0x3F)); *out += (char)(0x80 | ((cp >> 6) & 0x3F)); *out += (char)(0x80 | (cp & 0x3F)); }
                }
            } else {
                char r;
                switch (c) { case '"': r = '"'; break; case '\\': r = '\\'; break; case '/': r = '/'; break; case 'n': r = '\n'; break; case 't': r = '\t'; break;
                             case 'r': r = '\r'; break; case 'b': r = '\b'; break; case 'f': r = '\f'; break; default: return false; }   // (json.load: "Invalid \\escape")
                if (out) *out += r;
            }
        }
    }
    // a value that is neither string nor container, as json.load takes it: true / false / null, the three names Python's json adds (NaN, Infinity,
    // -Infinity), or a number -?(0|[1-9][0-9]*)(\.[0-9]+)?([eE][-+]?[0-9]+)? — anything else is the JSONDecodeError the reference dies with
    bool literal() {
        auto word = [&](const char *w) { const size_t n = strlen(w); if ((size_t)(e - p) >= n && !memcmp(p, w, n)) { p += n; return true; } return false; };
        if (word("true") || word("false") || word("null") || word("NaN") || word("Infinity") || word("-Infinity")) return true;
        if (p < e && *p == '-') ++p;
        if (p >= e || *p < '0' || *p > '9') return false;
        if (*p == '0') ++p; else while (p < e && *p >= '0' && *p <= '9') ++p;
        if (p < e && *p == '.') { const uint8_t *q = ++p; while (p < e && *p >= '0' && *p <= '9') ++p; if (p == q) return false; }
        if (p < e && (*p == 'e' || *p == 'E')) {
            const uint8_t *save = p;
            ++p;
            if (p < e && (*p == '+' || *p == '-')) ++p;
            const uint8_t *q = p;
            while (p < e && *p >= '0' && *p <= '9') ++p;
            if (p == q) p = save;                               // ("1e" / "1e+": the number ends in front of the 'e', what follows is the caller's error)
        }
        return true;
    }
    bool skip(int depth = 0) {                                  // any JSON value
        ws();
        if (p >= e || depth > 900) return false;                // (nesting that deep: Python's scanner dies with RecursionError)
        if (*p == '"') return str(nullptr);
        if (*p == '[') { ++p; if (eat(']')) return true; do { if (!skip(depth + 1)) return false; } while (eat(',')); return eat(']'); }
        if (*p == '{') { ++p; if (eat('}')) return true; do { if (!str(nullptr) || !eat(':') || !skip(depth + 1)) return false; } while (eat(',')); return eat('}'); }
        if (!literal()) return false;
        return p >= e || *p == ',' || *p == ']' || *p == '}' || *p == ' ' || *p == '\n' || *p == '\r' || *p == '\t';
    }
    bool count_list(uint64_t &n) {                              // '[' values ']' -> number of values
        n = 0;
        if (!eat('[')) return false;
        if (eat(']')) return true;
        do { if (!skip()) return false; ++n; } while (eat(','));
        return eat(']');
    }
    // one pair  "key": [list, list, ...]  -> the key (NUL terminated, appended) and the lengths of its first two lists
    bool pair(std::string &keys, std::vector<uint64_t> &cnt) {
        std::string k;
        uint64_t a = 0, b = 0;
        if (!str(&k) || !eat(':') || !eat('[') || !count_list(a) || !eat(',') || !count_list(b)) return false;
        while (eat(',')) if (!skip()) return false;             // further elements are never looked at
        if (!eat(']')) return false;
        if (k.find('\0') != std::string::npos) return false;
        keys += k; keys += '\0';
        cnt.push_back(a); cnt.push_back(b);
        return true;
    }
};

// the whole text by one thread: 0 or SVJG_E_INPUT
int count_sequential(const uint8_t *base, size_t n, std::string &keys, std::vector<uint64_t> &cnt) {
    Scan s{base, base + n};
    if (!s.eat('{')) return SVJG_E_INPUT;
    if (!s.eat('}')) {
        do { if (!s.pair(keys, cnt)) return SVJG_E_INPUT; } while (s.eat(','));
        if (!s.eat('}')) return SVJG_E_INPUT;
    }
    s.ws();
    return s.p == s.e ? 0 : SVJG_E_INPUT;
}

// several threads, chained (see above): true = keys / cnt hold the result; false = the chain did not hold (or the text is malformed):
// the caller parses sequentially, which also decides what a malformed text is
bool count_chained(const uint8_t *base, size_t n, int T, std::string &keys, std::vector<uint64_t> &cnt) {
    Scan s0{base, base + n};
    if (!s0.eat('{')) return false;
    s0.ws();
    if (s0.p >= s0.e || *s0.p != '"') return false;
    std::vector<const uint8_t *> start{s0.p};
    static const char pat[] = "\n    \"";
    for (int i = 1; i < T; ++i) {
        const uint8_t *from = base + (size_t)((double)n * i / T);
        if (from <= start.back()) continue;
        const void *hit = memmem(from, (size_t)(base + n - from), pat, 6);
        if (!hit) break;
        const uint8_t *q = (const uint8_t *)hit + 5;             // the quote
        if (q > start.back()) start.push_back(q);
    }
    const size_t P = start.size();
    if (P < 2) return false;
    std::vector<std::string> tk(P);
    std::vector<std::vector<uint64_t>> tc(P);
    std::vector<uint8_t> ok(P, 0);
    std::vector<std::thread> th;
    for (size_t i = 0; i < P; ++i)
        th.emplace_back([&, i] {
            Scan s{start[i], base + n};
            const uint8_t *limit = i + 1 < P ? start[i + 1] : nullptr;
            for (;;) {
                if (!s.pair(tk[i], tc[i])) return;
                if (limit) {
                    if (!s.eat(',')) return;                     // (the object ends here, or garbage: not where the next thread starts)
                    s.ws();
                    if (s.p == limit) { ok[i] = 1; return; }
                    if (s.p > limit) return;
                } else {
                    if (s.eat(',')) continue;
                    if (!s.eat('}')) return;
                    s.ws();
                    ok[i] = s.p == s.e;
                    return;
                }
            }
        });
    for (auto &x : th) x.join();
    for (size_t i = 0; i < P; ++i) if (!ok[i]) return false;
    for (size_t i = 0; i < P; ++i) { keys += tk[i]; cnt.insert(cnt.end(), tc[i].begin(), tc[i].end()); }
    return true;
}

}  // namespace

extern "C" void svjg_host_free(void *p) { free(p); }

// keys_out: malloc'd blob of n NUL-terminated keys in file order; counts_out: malloc'd uint64[n][2].
// Returns 0, SVJG_E_NOMEM (cannot read), or SVJG_E_INPUT (not the JSON shape predict-genotype.py needs).
extern "C" int svjg_count_informative_json(const char *path, char **keys_out, uint64_t *keys_len, uint64_t **counts_out, uint64_t *n_keys)
{
    if (!path || !keys_out || !keys_len || !counts_out || !n_keys) return SVJG_E_ARG;
    *keys_out = nullptr; *counts_out = nullptr; *keys_len = 0; *n_keys = 0;
    int fd = open(path, O_RDONLY);
    if (fd < 0) return SVJG_E_NOMEM;
    struct stat st;
    if (fstat(fd, &st)) { close(fd); return SVJG_E_NOMEM; }
    size_t n = (size_t)st.st_size;
    const uint8_t *base = n ? (const uint8_t *)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
    close(fd);
    if (n && base == MAP_FAILED) return SVJG_E_NOMEM;
    std::string keys; std::vector<uint64_t> cnt;
    int T = (int)std::thread::hardware_concurrency();
    { const char *e = getenv("SVJG_JSON_THREADS"); if (e && atoi(e) > 0) T = atoi(e); }
    if (T > 32) T = 32;
    size_t min_par = (size_t)64 << 20;
    { const char *e = getenv("SVJG_JSON_PARALLEL_FROM"); if (e) min_par = (size_t)strtoull(e, nullptr, 10); }   // (tests: a small file through the threads)
    int rc = 0;
    bool done = false;
    if (T > 1 && n >= min_par) done = count_chained(base, n, T, keys, cnt);
    if (!done) { keys.clear(); cnt.clear(); rc = count_sequential(base, n, keys, cnt); }
    if (n) munmap((void *)base, n);
    if (rc) return rc;
    *keys_out = (char *)malloc(keys.size() + 1);
    *counts_out = (uint64_t *)malloc((cnt.size() + 1) * sizeof(uint64_t));
    if (!*keys_out || !*counts_out) { free(*keys_out); free(*counts_out); return SVJG_E_NOMEM; }
    if (!keys.empty()) memcpy(*keys_out, keys.data(), keys.size());
    if (!cnt.empty()) memcpy(*counts_out, cnt.data(), cnt.size() * sizeof(uint64_t));
    *keys_len = keys.size(); *n_keys = cnt.size() / 2;
    return 0;
}
