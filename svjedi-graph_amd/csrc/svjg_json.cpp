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

inline void hex4(std::string &o, uint32_t v) {
    static const char *H = "0123456789abcdef";
    o += "\\u"; o += H[(v >> 12) & 15]; o += H[(v >> 8) & 15]; o += H[(v >> 4) & 15]; o += H[v & 15];
}

// JSON string body with ensure_ascii=True; returns false on malformed UTF-8
bool escape(std::string &o, const uint8_t *s, size_t n) {
    for (size_t i = 0; i < n;) {
        uint8_t c = s[i];
        if (c < 0x80) {
            switch (c) {
                case '"': o += "\\\""; break;
                case '\\': o += "\\\\"; break;
                case '\n': o += "\\n"; break;
                case '\r': o += "\\r"; break;
                case '\t': o += "\\t"; break;
                case '\b': o += "\\b"; break;
                case '\f': o += "\\f"; break;
                default: if (c < 0x20 || c == 0x7F) hex4(o, c); else o += (char)c;   // json escapes everything outside ' '..'~'
            }
            ++i;
            continue;
        }
        uint32_t cp; size_t len;
        if ((c & 0xE0) == 0xC0) { cp = c & 0x1F; len = 2; }
        else if ((c & 0xF0) == 0xE0) { cp = c & 0x0F; len = 3; }
        else if ((c & 0xF8) == 0xF0) { cp = c & 0x07; len = 4; }
        else return false;
        if (i + len > n) return false;
        for (size_t k = 1; k < len; ++k) { if ((s[i + k] & 0xC0) != 0x80) return false; cp = (cp << 6) | (s[i + k] & 0x3F); }
        if ((len == 2 && cp < 0x80) || (len == 3 && cp < 0x800) || (len == 4 && cp < 0x10000) || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) return false;
        if (cp >= 0x10000) { cp -= 0x10000; hex4(o, 0xD800 + (cp >> 10)); hex4(o, 0xDC00 + (cp & 0x3FF)); }
        else hex4(o, cp);
        i += len;
    }
    return true;
}

// escaped text of the line starting at `start`
bool line_text(const Job &j, uint64_t start, std::string &o) {
    uint64_t e = start;
    while (e < j.n && j.gaf[e] != '\n' && j.gaf[e] != '\r') ++e;
    // cut before the first "cg:Z:" (searched in the translated text: the tag cannot span the terminator)
    uint64_t cut = e; bool tag = false;
    for (uint64_t q = start; q + 5 <= e; ++q)
        if (j.gaf[q] == 'c' && j.gaf[q + 1] == 'g' && j.gaf[q + 2] == ':' && j.gaf[q + 3] == 'Z' && j.gaf[q + 4] == ':') { cut = q; tag = true; break; }
    if (!escape(o, j.gaf + start, (size_t)(cut - start))) return false;
    if (!tag && e < j.n) o += "\\n";
    return true;
}

bool render_key(const Job &j, uint64_t ki, std::string &o, std::string &tmp) {
    const uint32_t slot = j.order[ki];
    o += ki ? ",\n    \"" : "\n    \"";
    const char *id = j.sv_ids[slot];
    if (!escape(o, (const uint8_t *)id, strlen(id))) return false;
    o += "\": [";
    for (int allele = 0; allele < 2; ++allele) {
        uint64_t cnt = 0;
        for (uint64_t r = j.slot_begin[slot]; r < j.slot_begin[slot + 1]; ++r) {
            const uint32_t rep = allele ? j.recs[r].n_alt : j.recs[r].n_ref;
            if (!rep) continue;
            tmp.clear();
            if (!line_text(j, j.recs[r].line_start, tmp)) return false;
            for (uint32_t k = 0; k < rep; ++k) {
                o += cnt ? ",\n            \"" : "\n        [\n            \"";
                o += tmp; o += '"';
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
    // group by slot (counting sort), then file order inside each group
    std::vector<uint64_t> begin((size_t)n_slots + 1, 0);
    for (uint64_t i = 0; i < n_recs; ++i) { if (recs_in[i].slot >= n_slots) return SVJG_E_ARG; begin[recs_in[i].slot + 1]++; }
    for (uint32_t s = 0; s < n_slots; ++s) begin[s + 1] += begin[s];
    std::vector<svjg_hitrec> recs(n_recs);
    { std::vector<uint64_t> cur(begin.begin(), begin.end() - 1);
      for (uint64_t i = 0; i < n_recs; ++i) recs[cur[recs_in[i].slot]++] = recs_in[i]; }
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
    // in parallel; a task's file offset is known as soon as all earlier tasks are rendered, and the worker that rendered it
    // writes it there itself (pwrite): both the rendering and the copy into the page cache run on all threads.
    const uint64_t REC_PER_TASK = 16384;
    std::vector<uint64_t> task_lo{0};
    { uint64_t acc = 0;
      for (uint64_t ki = 0; ki < order.size(); ++ki) {
          acc += begin[order[ki] + 1] - begin[order[ki]];
          if (acc >= REC_PER_TASK && ki + 1 < order.size()) { task_lo.push_back(ki + 1); acc = 0; }
      }
      task_lo.push_back(order.size()); }
    const uint64_t n_tasks = task_lo.size() - 1;
    int T = n_threads > 0 ? n_threads : (int)std::min(64u, std::thread::hardware_concurrency());
    if (T < 1) T = 1;
    if ((uint64_t)T > n_tasks) T = (int)n_tasks;
    std::vector<uint64_t> off(n_tasks + 1, 0);                 // off[t] = file offset of task t, valid once t <= known
    std::vector<uint64_t> size(n_tasks, 0);
    std::vector<char> done(n_tasks, 0);
    off[0] = 1;                                                // behind the opening brace
    std::atomic<uint64_t> next{0};
    std::atomic<int> bad{0};                                   // 1 = text is not valid UTF-8, 2 = write error
    std::mutex mu; std::condition_variable cv;
    uint64_t known = 0;                                        // tasks 0 .. known-1 are rendered (workers stay <= 4T ahead: bounds memory)
    auto worker = [&]() {
        std::string tmp, s;
        for (;;) {
            const uint64_t t = next.fetch_add(1);
            if (t >= n_tasks) return;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return t < known + 4 * (uint64_t)T || bad.load(); }); }
            if (bad.load()) return;
            s.clear();
            for (uint64_t ki = task_lo[t]; ki < task_lo[t + 1]; ++ki) {
                const uint32_t slot = order[ki];
                // every line of every group sorted by file order
                std::sort(recs.begin() + begin[slot], recs.begin() + begin[slot + 1],
                          [](const svjg_hitrec &a, const svjg_hitrec &b) { return a.line_start < b.line_start; });
                if (!render_key(job, ki, s, tmp)) { bad.store(1); break; }
            }
            uint64_t at;
            { std::unique_lock<std::mutex> lk(mu);
              size[t] = s.size(); done[t] = 1;
              while (known < n_tasks && done[known]) { off[known + 1] = off[known] + size[known]; ++known; }
              cv.notify_all();
              cv.wait(lk, [&] { return known >= t || bad.load(); });
              at = off[t]; }
            if (bad.load()) { cv.notify_all(); return; }
            if (!put(s.data(), s.size(), at)) { bad.store(2); cv.notify_all(); return; }
        }
    };
    int rc = 0;
    if (!put("{", 1, 0)) rc = SVJG_E_NOMEM;
    const auto t_mid = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    if (!rc) for (int i = 0; i < T; ++i) th.emplace_back(worker);
    for (auto &x : th) x.join();
    if (!rc && bad.load()) rc = bad.load() == 1 ? SVJG_E_INPUT : SVJG_E_NOMEM;
    if (!rc && !put("\n}", 2, off[n_tasks])) rc = SVJG_E_NOMEM;
    if (close(fd) && !rc) rc = SVJG_E_NOMEM;
    if (verbose) {
        const auto t_end = std::chrono::steady_clock::now();
        fprintf(stderr, "[svjg] json writer: group + sort %.2f s, render + write %.2f s (%d threads, %llu tasks)\n",
                std::chrono::duration<double>(t_mid - t_start).count(), std::chrono::duration<double>(t_end - t_mid).count(), T, (unsigned long long)n_tasks);
    }
    return rc;
}

// ---------------------------------------------------------------------------------------------------------------
// Reader side: predict-genotype.py only needs len(dict[key][0]) and len(dict[key][1]) (predict-genotype.py:219-226).
// svjg_count_informative_json scans the JSON text once (any valid JSON of that shape, not only json.dumps' layout)
// and returns the keys (unescaped, UTF-8, NUL separated) and the two list lengths per key.
// ---------------------------------------------------------------------------------------------------------------

namespace {

struct Scan {
    const uint8_t *p, *e;
    bool fail = false;
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p; }
    bool eat(char c) { ws(); if (p < e && *p == (uint8_t)c) { ++p; return true; } return false; }
    // string body -> out (unescaped UTF-8) or skipped when out == nullptr
    bool str(std::string *out) {
        ws();
        if (p >= e || *p != '"') return false;
        ++p;
        while (p < e && *p != '"') {
            if (*p == '\\') {
                if (p + 1 >= e) return false;
                uint8_t c = p[1];
                p += 2;
                if (c == 'u') {
                    if (p + 4 > e) return false;
                    auto hx = [&](const uint8_t *q, uint32_t &v) { v = 0; for (int i = 0; i < 4; ++i) { uint8_t h = q[i]; v <<= 4;
                        if (h >= '0' && h <= '9') v |= h - '0'; else if (h >= 'a' && h <= 'f') v |= h - 'a' + 10; else if (h >= 'A' && h <= 'F') v |= h - 'A' + 10; else return false; } return true; };
                    uint32_t cp;
                    if (!hx(p, cp)) return false;
                    p += 4;
                    if (cp >= 0xD800 && cp <= 0xDBFF && p + 6 <= e && p[0] == '\\' && p[1] == 'u') {
                        uint32_t lo;
                        if (hx(p + 2, lo) && lo >= 0xDC00 && lo <= 0xDFFF) { cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00); p += 6; }
                    }
                    if (out) {
                        if (cp < 0x80) *out += (char)cp;
                        else if (cp < 0x800) { *out += (char)(0xC0 | (cp >> 6)); *out += (char)(0x80 | (cp & 0x3F)); }
                        else if (cp < 0x10000) { *out += (char)(0xE0 | (cp >> 12)); *out += (char)(0x80 | ((cp >> 6) & 0x3F)); *out += (char)(0x80 | (cp & 0x3F)); }
                        else { *out += (char)(0xF0 | (cp >> 18)); *out += (char)(0x80 | ((cp >> 12) & 0x3F)); *out += (char)(0x80 | ((cp >> 6) & 0x3F)); *out += (char)(0x80 | (cp & 0x3F)); }
                    }
                } else if (out) {
                    switch (c) { case 'n': *out += '\n'; break; case 't': *out += '\t'; break; case 'r': *out += '\r'; break;
                                 case 'b': *out += '\b'; break; case 'f': *out += '\f'; break; default: *out += (char)c; }
                }
            } else {
                if (out) *out += (char)*p;
                ++p;
            }
        }
        if (p >= e) return false;
        ++p;
        return true;
    }
    bool skip() {                                               // any JSON value
        ws();
        if (p >= e) return false;
        if (*p == '"') return str(nullptr);
        if (*p == '[') { ++p; if (eat(']')) return true; do { if (!skip()) return false; } while (eat(',')); return eat(']'); }
        if (*p == '{') { ++p; if (eat('}')) return true; do { if (!str(nullptr) || !eat(':') || !skip()) return false; } while (eat(',')); return eat('}'); }
        const uint8_t *q = p;
        while (p < e && *p != ',' && *p != ']' && *p != '}' && *p != ' ' && *p != '\n' && *p != '\r' && *p != '\t') ++p;
        return p > q;
    }
    bool count_list(uint64_t &n) {                              // '[' values ']' -> number of values
        n = 0;
        if (!eat('[')) return false;
        if (eat(']')) return true;
        do { if (!skip()) return false; ++n; } while (eat(','));
        return eat(']');
    }
};

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
    Scan s{base, base + n};
    std::string keys; std::vector<uint64_t> cnt;
    int rc = 0;
    if (!s.eat('{')) rc = SVJG_E_INPUT;
    else if (!s.eat('}')) {
        do {
            std::string k;
            uint64_t a = 0, b = 0;
            if (!s.str(&k) || !s.eat(':') || !s.eat('[') || !s.count_list(a) || !s.eat(',') || !s.count_list(b)) { rc = SVJG_E_INPUT; break; }
            while (s.eat(',')) if (!s.skip()) { rc = SVJG_E_INPUT; break; }     // further elements are never looked at
            if (rc || !s.eat(']')) { rc = SVJG_E_INPUT; break; }
            if (k.find('\0') != std::string::npos) { rc = SVJG_E_INPUT; break; }
            keys += k; keys += '\0';
            cnt.push_back(a); cnt.push_back(b);
        } while (s.eat(','));
        if (!rc && !s.eat('}')) rc = SVJG_E_INPUT;
    }
    if (!rc) { s.ws(); if (s.p != s.e) rc = SVJG_E_INPUT; }
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
