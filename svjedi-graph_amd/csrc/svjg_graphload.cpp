// libsvjg_host.so — native loader of the two graph inputs of the alignment filter (no GPU code in this file):
//   <prefix>_svs_edges.json : link key "L@s@R@s" -> [[sv_id, allele], ...]      (filter-alignments.py:95-98)
//   the alt-node S-lines of the GFA : name -> sequence length                     (filter-alignments.py:103-113)
// -> the flat tables of include/svjg.h (sorted node table, CSR of directed links with the forward and reversed
// dictionary entries pre-merged, chromosome dictionary, count-slot numbering, hazard flags).
//
// This is the FAST path of svjg/graph.py for the files construct-graph.py writes (plain ASCII, json.dumps layout or any
// other whitespace, canonical node names).  svjg/graph.py holds the semantics: anything this parser does not recognise
// — escapes or non-ASCII bytes in strings, duplicate keys, odd node names, an allele that is not 0 or 1, carriage returns
// in the GFA, ... — makes it return SVJG_E_UNSUPPORTED and the Python loader takes over (and raises what it raises).
// tests/test_graph_native.py checks that both give identical tables.
#include "../../include/svjg.h"
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

struct Mapped {
    const uint8_t *p = nullptr; size_t n = 0; int fd = -1;
    bool open_(const char *path) {
        fd = open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st)) return false;
        n = (size_t)st.st_size;
        if (n == 0) { p = (const uint8_t *)""; return true; }
        void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return false;
        p = (const uint8_t *)m;
        return true;
    }
    ~Mapped() { if (p && n) munmap((void *)p, n); if (fd >= 0) close(fd); }
};

struct Str { const uint8_t *p; uint32_t n; };
struct StrHash { size_t operator()(const Str &s) const { uint64_t h = 1469598103934665603ull; for (uint32_t i = 0; i < s.n; ++i) h = (h ^ s.p[i]) * 1099511628211ull; return (size_t)h; } };
struct StrEq { bool operator()(const Str &a, const Str &b) const { return a.n == b.n && !memcmp(a.p, b.p, a.n); } };

// Str -> uint32 with open addressing (the tables hold millions of names: a node-based std::unordered_map spends its time in
// malloc and pointer chasing).  Keys keep their insertion order in `keys`.
struct StrMap {
    std::vector<Str> keys; std::vector<uint32_t> vals;
    std::vector<uint32_t> slot, tag;                                  // slot[i] = index into keys + 1 (0 = empty), tag[i] = hash bits
    size_t mask = 0;
    static uint64_t hash(const Str &s) {
        uint64_t h = 0x9E3779B97F4A7C15ull ^ s.n;
        const uint8_t *p = s.p; uint32_t n = s.n;
        for (; n >= 8; p += 8, n -= 8) { uint64_t w; memcpy(&w, p, 8); h = (h ^ w) * 0xFF51AFD7ED558CCDull; h ^= h >> 32; }
        if (n) { uint64_t w = 0; memcpy(&w, p, n); h = (h ^ w) * 0xFF51AFD7ED558CCDull; h ^= h >> 32; }
        h *= 0xC4CEB9FE1A85EC53ull;
        return h ^ (h >> 29);
    }
    void grow() {
        const size_t cap = mask ? (mask + 1) * 2 : 1024;
        slot.assign(cap, 0); tag.assign(cap, 0); mask = cap - 1;
        for (size_t k = 0; k < keys.size(); ++k) {
            const uint64_t h = hash(keys[k]);
            size_t i = (size_t)h & mask;
            while (slot[i]) i = (i + 1) & mask;
            slot[i] = (uint32_t)k + 1; tag[i] = (uint32_t)(h >> 32);
        }
    }
    uint32_t *find(const Str &s) {
        if (!mask) return nullptr;
        const uint64_t h = hash(s);
        for (size_t i = (size_t)h & mask; slot[i]; i = (i + 1) & mask)
            if (tag[i] == (uint32_t)(h >> 32)) { const Str &k = keys[slot[i] - 1]; if (k.n == s.n && !memcmp(k.p, s.p, s.n)) return &vals[slot[i] - 1]; }
        return nullptr;
    }
    // the value of s, inserted as v if s is new; `fresh` tells which
    uint32_t &put(const Str &s, uint32_t v, bool *fresh = nullptr) {
        if (uint32_t *q = find(s)) { if (fresh) *fresh = false; return *q; }
        if ((keys.size() + 1) * 2 > mask + 1) grow();
        const uint64_t h = hash(s);
        size_t i = (size_t)h & mask;
        while (slot[i]) i = (i + 1) & mask;
        keys.push_back(s); vals.push_back(v);
        slot[i] = (uint32_t)keys.size(); tag[i] = (uint32_t)(h >> 32);
        if (fresh) *fresh = true;
        return vals.back();
    }
    size_t size() const { return keys.size(); }
};

inline bool py_ws(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 28 && c <= 31); }

// canonical unsigned decimal (svjg/graph.py: _canon_uint): digits, no leading zero, <= 10 digits, <= 0xFFFFFFFF
inline bool canon_uint(const uint8_t *p, uint32_t n, uint32_t &v) {
    if (n == 0 || n > 10 || (n > 1 && p[0] == '0')) return false;
    uint64_t x = 0;
    for (uint32_t i = 0; i < n; ++i) { if (p[i] < '0' || p[i] > '9') return false; x = x * 10 + (p[i] - '0'); }
    if (x > 0xFFFFFFFFull) return false;
    v = (uint32_t)x;
    return true;
}

struct NodeInfo { Str name; Str chrom; uint32_t pos, kind, v2, cidx; uint64_t key; uint32_t aux; };

// 'chrom:start-end' -> (chrom, start, 0, end) ; 'chrom:pos.cnt' -> (chrom, pos, 1, cnt)   (svjg/graph.py: parse_node_name)
inline bool parse_name(Str nm, NodeInfo &o) {
    int colon = -1;
    for (int i = (int)nm.n - 1; i >= 0; --i) if (nm.p[i] == ':') { colon = i; break; }
    if (colon < 0) return false;
    const uint8_t *c = nm.p + colon + 1; const uint32_t cn = nm.n - (uint32_t)colon - 1;
    int sep = -1; uint32_t kind = 0;
    for (uint32_t i = 0; i < cn; ++i) if (c[i] == '-') { sep = (int)i; break; }
    if (sep < 0) { kind = 1; for (uint32_t i = 0; i < cn; ++i) if (c[i] == '.') { sep = (int)i; break; } }
    if (sep < 0) return false;
    uint32_t a, b;
    if (!canon_uint(c, (uint32_t)sep, a) || !canon_uint(c + sep + 1, cn - (uint32_t)sep - 1, b)) return false;
    if (kind == 1 && b >= 32768) return false;
    if (kind == 0 && b < a) return false;
    o.name = nm; o.chrom = Str{nm.p, (uint32_t)colon}; o.pos = a; o.kind = kind; o.v2 = b;
    return true;
}

struct JsonKey { Str k, l, r; uint8_t sl, sr; uint32_t ent_lo, ent_hi; uint32_t li, ri; };   // li, ri: index of the two names in `info`   // entries [ent_lo, ent_hi) of `ents`
struct Ent { Str sv; uint8_t allele; };

struct Parser {
    const uint8_t *p, *e;
    void ws() {                                                       // (json.dumps(indent=4) output is half blanks: eight at a time)
        for (;;) {
            while (p + 8 <= e) { uint64_t w; memcpy(&w, p, 8); if (w != 0x2020202020202020ull) break; p += 8; }
            if (p < e && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p; else return;
        }
    }
    bool eat(char c) { ws(); if (p < e && *p == (uint8_t)c) { ++p; return true; } return false; }
    // plain string: printable ASCII without '"' and '\\'
    bool str(Str &s) {
        ws();
        if (p >= e || *p != '"') return false;
        const uint8_t *b = ++p;
        const uint8_t *q = (const uint8_t *)memchr(b, '"', (size_t)(e - b));
        if (!q) return false;
        uint8_t bad = 0;
        for (const uint8_t *c = b; c < q; ++c) bad |= (uint8_t)((*c == '\\') | (*c < 0x20) | (*c >= 0x7F));
        if (bad) return false;
        s = Str{b, (uint32_t)(q - b)};
        p = q + 1;
        return true;
    }
};

}  // namespace


struct svjg_hostgraph {
    std::vector<svjg_node> nodes;
    std::vector<svjg_edge> edges;
    std::vector<uint32_t> hits;
    std::string chrom_names; std::vector<uint32_t> chrom_off, chrom_lo;
    std::string sv_blob; uint32_t n_slots = 0, n_hazard = 0;
    svjg_graph view{};
};

extern "C" void svjg_graph_free(svjg_hostgraph *g) { delete g; }
extern "C" const svjg_graph *svjg_graph_view(const svjg_hostgraph *g) { return g ? &g->view : nullptr; }
extern "C" int svjg_graph_info(const svjg_hostgraph *g, const char **sv_blob, uint64_t *sv_blob_len, uint32_t *n_hazard) {
    if (!g) return SVJG_E_ARG;
    if (sv_blob) *sv_blob = g->sv_blob.data();
    if (sv_blob_len) *sv_blob_len = g->sv_blob.size();
    if (n_hazard) *n_hazard = g->n_hazard;
    return 0;
}

extern "C" int svjg_graph_load(const char *edges_json, const char *gfa_path, svjg_hostgraph **out) {
    if (!edges_json || !gfa_path || !out) return SVJG_E_ARG;
    *out = nullptr;
    Mapped js, gf;
    if (!js.open_(edges_json) || !gf.open_(gfa_path)) return SVJG_E_NOMEM;
    const bool verbose = getenv("SVJG_VERBOSE") != nullptr;          // stage timers on stderr (measurement only)
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!verbose) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[svjg] graph loader, %s: %.2f s\n", what, std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };

    // ---- alt node name -> len(sequence) from the GFA (filter-alignments.py:105-113) ---------------------------------------
    StrMap alt_len;
    // (on a thread of its own, beside the parse of the edge table)
    auto scan_gfa = [&]() -> int {
        if (gf.n && memchr(gf.p, '\r', gf.n)) return SVJG_E_UNSUPPORTED;          // Python's universal newlines would cut lines there
        {
            const uint8_t *p = gf.p, *e = gf.p + gf.n;
            while (p < e) {
                const uint8_t *nl = (const uint8_t *)memchr(p, '\n', (size_t)(e - p));
                const uint8_t *le = nl ? nl : e;
                if (*p == 'S') {
                    const uint8_t *t1 = (const uint8_t *)memchr(p, '\t', (size_t)(le - p));
                    if (!t1) return SVJG_E_UNSUPPORTED;
                    const uint8_t *t2 = (const uint8_t *)memchr(t1 + 1, '\t', (size_t)(le - t1 - 1));
                    if (!t2) return SVJG_E_UNSUPPORTED;
                    Str name{t1 + 1, (uint32_t)(t2 - t1 - 1)};
                    int colon = -1;
                    for (int i = (int)name.n - 1; i >= 0; --i) if (name.p[i] == ':') { colon = i; break; }
                    bool dot = false;
                    for (uint32_t i = (uint32_t)(colon + 1); i < name.n; ++i) dot |= name.p[i] == '.';
                    if (dot) {
                        const uint8_t *re = le;                                   // line.rstrip()
                        while (re > p && py_ws(re[-1])) --re;
                        if (re <= t2) return SVJG_E_UNSUPPORTED;                  // third column gone: IndexError in the reference
                        const uint8_t *t3 = (const uint8_t *)memchr(t2 + 1, '\t', (size_t)(re - t2 - 1));
                        const uint8_t *se = t3 ? t3 : re;
                        for (const uint8_t *q = p; q < se; ++q) if (*q >= 0x80) return SVJG_E_UNSUPPORTED;
                        if ((uint64_t)(se - t2 - 1) >= 0xFFFFFFFFull) return SVJG_E_UNSUPPORTED;
                        alt_len.put(name, 0) = (uint32_t)(se - t2 - 1);
                    }
                }
                if (!nl) break;
                p = nl + 1;
            }
        }

        return 0;
    };
    int gfa_rc = 0;
    std::thread gfa_thread([&] { gfa_rc = scan_gfa(); });
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{gfa_thread};
    // ---- the edge table -------------------------------------------------------------------------------------------------
    std::vector<JsonKey> keys;
    std::vector<Ent> ents;
    keys.reserve(js.n / 128 + 16); ents.reserve(js.n / 128 + 16);     // (indent=4 output: ~145 bytes per key)
    {
        Parser ps{js.p, js.p + js.n};
        if (!ps.eat('{')) return SVJG_E_UNSUPPORTED;
        // a duplicate key makes json.load keep the last one.  Two keys WITH entries that are equal give two equal forward
        // queries in the link table below and are refused there; a key with an empty list is remembered here and must not
        // repeat any other key.
        std::vector<Str> empties;
        if (!ps.eat('}')) {
            for (;;) {
                Str k;
                if (!ps.str(k) || !ps.eat(':') || !ps.eat('[')) return SVJG_E_UNSUPPORTED;
                JsonKey jk{}; jk.k = k; jk.ent_lo = (uint32_t)ents.size();
                if (!ps.eat(']')) {
                    for (;;) {
                        Ent en{};
                        if (!ps.eat('[') || !ps.str(en.sv) || !ps.eat(',')) return SVJG_E_UNSUPPORTED;
                        ps.ws();
                        if (ps.p >= ps.e || (*ps.p != '0' && *ps.p != '1')) return SVJG_E_UNSUPPORTED;
                        en.allele = (uint8_t)(*ps.p - '0'); ++ps.p;
                        if (ps.p < ps.e && ((*ps.p >= '0' && *ps.p <= '9') || *ps.p == '.' || *ps.p == 'e' || *ps.p == 'E')) return SVJG_E_UNSUPPORTED;
                        if (!ps.eat(']')) return SVJG_E_UNSUPPORTED;
                        ents.push_back(en);
                        if (ps.eat(',')) continue;
                        if (ps.eat(']')) break;
                        return SVJG_E_UNSUPPORTED;
                    }
                }
                jk.ent_hi = (uint32_t)ents.size();
                if (jk.ent_hi > jk.ent_lo) {                                          // a present-but-empty key contributes nothing
                    // "L@s@R@s"
                    const uint8_t *a1 = (const uint8_t *)memchr(k.p, '@', k.n);
                    if (!a1 || a1 + 2 >= k.p + k.n || a1[2] != '@' || (a1[1] != '+' && a1[1] != '-')) return SVJG_E_UNSUPPORTED;
                    const uint8_t *rb = a1 + 3, *ke = k.p + k.n;
                    const uint8_t *a3 = (const uint8_t *)memchr(rb, '@', (size_t)(ke - rb));
                    if (!a3 || a3 + 2 != ke || (a3[1] != '+' && a3[1] != '-')) return SVJG_E_UNSUPPORTED;
                    jk.l = Str{k.p, (uint32_t)(a1 - k.p)}; jk.sl = a1[1] == '-';
                    jk.r = Str{rb, (uint32_t)(a3 - rb)}; jk.sr = a3[1] == '-';
                    keys.push_back(jk);
                } else empties.push_back(k);
                if (ps.eat(',')) continue;
                if (ps.eat('}')) break;
                return SVJG_E_UNSUPPORTED;
            }
        }
        ps.ws();
        if (ps.p != ps.e) return SVJG_E_UNSUPPORTED;
        if (!empties.empty()) {
            StrMap em;
            for (auto &k : empties) em.put(k, 1);
            for (auto &jk : keys) if (em.find(jk.k)) return SVJG_E_UNSUPPORTED;
        }
    }

    lap("edge table JSON");
    gfa_thread.join();
    if (gfa_rc) return gfa_rc;
    lap("waiting for the alt node lengths from the GFA");
    // ---- nodes ------------------------------------------------------------------------------------------------------------
    StrMap name_ix;                                                   // name -> index into info
    std::vector<NodeInfo> info;
    auto add_name = [&](Str nm, uint32_t &ix) -> bool {
        if (const uint32_t *q = name_ix.find(nm)) { ix = *q; return true; }
        NodeInfo ni{};
        if (!parse_name(nm, ni)) return false;
        ix = (uint32_t)info.size();
        name_ix.put(nm, ix);
        info.push_back(ni);
        return true;
    };
    { uint32_t ix; for (size_t i = 0; i < alt_len.keys.size(); ++i) if (!add_name(alt_len.keys[i], ix)) return SVJG_E_UNSUPPORTED; }
    for (auto &k : keys) if (!add_name(k.l, k.li) || !add_name(k.r, k.ri)) return SVJG_E_UNSUPPORTED;
    // chromosomes sorted by their bytes
    std::vector<Str> chroms;
    {
        std::unordered_map<Str, int, StrHash, StrEq> cs;
        for (auto &ni : info) if (cs.emplace(ni.chrom, 1).second) chroms.push_back(ni.chrom);
        std::sort(chroms.begin(), chroms.end(), [](const Str &a, const Str &b) {
            int c = memcmp(a.p, b.p, std::min(a.n, b.n)); return c ? c < 0 : a.n < b.n; });
        if (chroms.size() >= 65535) return SVJG_E_UNSUPPORTED;
        std::unordered_map<Str, uint32_t, StrHash, StrEq> cidx;
        for (uint32_t i = 0; i < chroms.size(); ++i) cidx.emplace(chroms[i], i);
        for (auto &ni : info) {
            ni.cidx = cidx[ni.chrom];
            ni.key = ((uint64_t)ni.cidx << 48) | ((uint64_t)ni.pos << 16) | ((uint64_t)ni.kind << 15) | (ni.kind ? ni.v2 : 0u);
            if (ni.kind) { const uint32_t *it = alt_len.find(ni.name); ni.aux = it ? *it : SVJG_LEN_UNKNOWN; }
            else ni.aux = ni.v2;
        }
    }
    lap("node names");
    const uint32_t n_nodes = (uint32_t)info.size();
    std::vector<uint32_t> order(n_nodes);
    for (uint32_t i = 0; i < n_nodes; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return info[a].key < info[b].key; });
    for (uint32_t i = 1; i < n_nodes; ++i) if (info[order[i]].key == info[order[i - 1]].key) return SVJG_E_UNSUPPORTED;   // two nodes at one coordinate
    std::vector<uint32_t> id_of(n_nodes);                            // info index -> node id
    for (uint32_t i = 0; i < n_nodes; ++i) id_of[order[i]] = i;

    lap("node order");
    svjg_hostgraph *G = new svjg_hostgraph();
    G->nodes.resize((size_t)n_nodes + 1);
    for (uint32_t i = 0; i < n_nodes; ++i) { G->nodes[i].key = info[order[i]].key; G->nodes[i].aux = info[order[i]].aux; G->nodes[i].row = 0; }
    G->nodes[n_nodes].key = ~0ull; G->nodes[n_nodes].aux = 0; G->nodes[n_nodes].row = 0;
    G->chrom_off.assign(chroms.size() + 1, 0);
    for (size_t i = 0; i < chroms.size(); ++i) { G->chrom_names.append((const char *)chroms[i].p, chroms[i].n); G->chrom_off[i + 1] = (uint32_t)G->chrom_names.size(); }
    G->chrom_names.append(4, '\0');
    G->chrom_lo.assign(chroms.size() + 1, n_nodes);
    { uint32_t c = 0;
      for (uint32_t i = 0; i < n_nodes; ++i) { const uint32_t ci = (uint32_t)(G->nodes[i].key >> 48); while (c <= ci) G->chrom_lo[c++] = i; }
      while (c <= chroms.size()) G->chrom_lo[c++] = n_nodes; }

    // ---- directed link table: T'[q] = d[q] ++ d[reverse(q)], sorted by (left, strand, right, strand) ---------------------------
    struct Q { uint64_t key; uint32_t jk; uint8_t rev; };          // key = left << 33 | sl << 32 | right << 1 | sr
    std::vector<Q> qs;
    qs.reserve(keys.size() * 2);
    for (uint32_t i = 0; i < keys.size(); ++i) {
        const uint32_t a = id_of[keys[i].li], b = id_of[keys[i].ri];
        const uint32_t sl = keys[i].sl, sr = keys[i].sr;
        qs.push_back(Q{((uint64_t)a << 33) | ((uint64_t)sl << 32) | ((uint64_t)b << 1) | sr, i, 0});
        qs.push_back(Q{((uint64_t)b << 33) | ((uint64_t)(sr ^ 1u) << 32) | ((uint64_t)a << 1) | (sl ^ 1u), i, 1});
    }
    std::sort(qs.begin(), qs.end(), [](const Q &x, const Q &y) { return x.key != y.key ? x.key < y.key : x.rev < y.rev; });
    StrMap slot_of;
    std::vector<uint32_t> left_count((size_t)n_nodes + 1, 0);
    std::vector<uint32_t> hv;
    G->edges.reserve(qs.size());
    for (size_t i = 0; i < qs.size();) {
        size_t j = i;
        while (j < qs.size() && qs[j].key == qs[i].key) ++j;
        // at most one forward and one reversed entry per query (keys are unique and name <-> id is one to one)
        if (j - i > 2 || (j - i == 2 && qs[i].rev == qs[i + 1].rev)) { delete G; return SVJG_E_UNSUPPORTED; }
        hv.clear();
        for (size_t t = i; t < j; ++t)
            for (uint32_t en = keys[qs[t].jk].ent_lo; en < keys[qs[t].jk].ent_hi; ++en) {
                bool fresh;
                const uint32_t s = slot_of.put(ents[en].sv, (uint32_t)slot_of.size(), &fresh);
                if (fresh) { G->sv_blob.append((const char *)ents[en].sv.p, ents[en].sv.n); G->sv_blob.push_back('\0'); }
                hv.push_back((s << 1) | ents[en].allele);
            }
        svjg_edge ed{};
        const uint32_t a = (uint32_t)(qs[i].key >> 33);
        ed.right = (uint32_t)(qs[i].key >> 1) & 0x7FFFFFFFu;
        ed.meta = (uint32_t)((qs[i].key >> 32) & 1u) | ((uint32_t)(qs[i].key & 1u) << 1) | ((uint32_t)hv.size() << 2);
        if (hv.size() <= 2) { ed.h0 = hv[0]; ed.h1 = hv.size() > 1 ? hv[1] : 0; }
        else { ed.h0 = (uint32_t)G->hits.size(); ed.h1 = 0; G->hits.insert(G->hits.end(), hv.begin(), hv.end()); }
        G->edges.push_back(ed);
        left_count[a + 1]++;
        i = j;
    }
    if (slot_of.size() >= (1u << 30)) { delete G; return SVJG_E_UNSUPPORTED; }
    G->n_slots = (uint32_t)slot_of.size();
    for (uint32_t i = 0; i < n_nodes; ++i) left_count[i + 1] += left_count[i];
    for (uint32_t i = 0; i <= n_nodes; ++i) G->nodes[i].row = left_count[i];

    lap("link table");
    // ---- names that are proper substrings of other names (strand quirk, filter-alignments.py:206; svjg/graph.py: _hazards) -----
    {
        bool colon_in_chrom = false;
        for (auto &c : chroms) if (memchr(c.p, ':', c.n)) colon_in_chrom = true;
        std::vector<uint8_t> hz(n_nodes, 0);
        if (colon_in_chrom) {
            // any contig names (svjg/graph.py: _hazards_general): wherever a node name X = Cx:Tx occurs inside a name Y, X's LAST colon
            // meets one of Y's colons — Cx is a suffix of what stands in front of that colon, Tx a prefix of what follows it.  For every Y
            // and every colon of Y: the contigs that end there x the prefixes behind it that spell a tail (digits, '-' or '.', digits).
            std::string cand;
            // (r06: the contigs that END at a colon are looked up by their distinct lengths in a set — thousands of contigs x every colon of
            //  every node name was seconds here and minutes in the Python loader for an analysis set)
            std::unordered_set<std::string> cset;
            std::vector<uint32_t> clens;
            for (auto &c : chroms) { cset.emplace((const char *)c.p, c.n); clens.push_back(c.n); }
            std::sort(clens.begin(), clens.end());
            clens.erase(std::unique(clens.begin(), clens.end()), clens.end());
            std::string probe;
            for (uint32_t y = 0; y < n_nodes; ++y) {
                const Str Y = info[y].name;
                for (uint32_t c = 0; c < Y.n; ++c) {
                    if (Y.p[c] != ':') continue;
                    const uint8_t *rest = Y.p + c + 1; const uint32_t rn = Y.n - c - 1;
                    uint32_t i = 0;
                    while (i < rn && rest[i] >= '0' && rest[i] <= '9') ++i;
                    if (i == 0 || i >= rn || (rest[i] != '-' && rest[i] != '.')) continue;
                    for (const uint32_t cn : clens) {
                        if (cn > c) break;
                        probe.assign((const char *)Y.p + c - cn, cn);
                        if (!cset.count(probe)) continue;
                        const Str cx{Y.p + c - cn, cn};
                        for (uint32_t j = i + 1; j < rn && rest[j] >= '0' && rest[j] <= '9'; ++j) {
                            cand.assign((const char *)cx.p, cx.n); cand.push_back(':'); cand.append((const char *)rest, j + 1);
                            const Str X{(const uint8_t *)cand.data(), (uint32_t)cand.size()};
                            if (X.n == Y.n && !memcmp(X.p, Y.p, X.n)) continue;
                            if (uint32_t *q = name_ix.find(X)) hz[*q] = 1;
                        }
                    }
                }
            }
        } else {
            // suffix[c][d] : chromosome d ends with chromosome c
            const size_t nc = chroms.size();
            std::vector<std::vector<uint32_t>> suffix_of(nc);
            for (size_t c = 0; c < nc; ++c)
                for (size_t d = 0; d < nc; ++d)
                    if (chroms[d].n >= chroms[c].n && !memcmp(chroms[d].p + chroms[d].n - chroms[c].n, chroms[c].p, chroms[c].n)) suffix_of[c].push_back((uint32_t)d);
            // groups of nodes with one (pos, kind)
            std::vector<uint32_t> idx(n_nodes);
            for (uint32_t i = 0; i < n_nodes; ++i) idx[i] = i;
            auto gkey = [&](uint32_t i) { return ((uint64_t)info[i].pos << 1) | info[i].kind; };
            std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return gkey(a) < gkey(b); });
            auto tail_of = [&](uint32_t i) {                       // str(v2): the text behind the separator (canonical decimal)
                const NodeInfo &ni = info[i];
                const uint8_t *e = ni.name.p + ni.name.n, *q = e;
                while (q > ni.name.p && q[-1] >= '0' && q[-1] <= '9') --q;
                return Str{q, (uint32_t)(e - q)};
            };
            for (size_t i = 0; i < idx.size();) {
                size_t j = i;
                while (j < idx.size() && gkey(idx[j]) == gkey(idx[i])) ++j;
                if (j - i >= 2)
                    for (size_t x = i; x < j; ++x) {
                        const uint32_t n = idx[x];
                        const Str t = tail_of(n);
                        for (uint32_t d : suffix_of[info[n].cidx])
                            for (size_t y = i; y < j; ++y) {
                                const uint32_t n2 = idx[y];
                                if (n2 == n || info[n2].cidx != d) continue;
                                const Str t2 = tail_of(n2);
                                if (t2.n >= t.n && !memcmp(t2.p, t.p, t.n)) hz[n] = 1;
                            }
                    }
                i = j;
            }
        }
        for (uint32_t i = 0; i < n_nodes; ++i) if (hz[i]) { G->nodes[id_of[i]].row |= 0x80000000u; ++G->n_hazard; }
    }

    lap("substring hazards");
    const uint64_t n_edges = G->edges.size(), n_hits = G->hits.size();
    if (G->edges.empty()) G->edges.push_back(svjg_edge{});           // (the arrays are never empty, like numpy's in svjg/graph.py)
    if (G->hits.empty()) G->hits.push_back(0);
    G->view.nodes = G->nodes.data(); G->view.n_nodes = n_nodes;
    G->view.edges = G->edges.data(); G->view.n_edges = n_edges;
    G->view.hits = G->hits.data(); G->view.n_hits = n_hits;
    G->view.chrom_names = G->chrom_names.data(); G->view.chrom_off = G->chrom_off.data(); G->view.chrom_node_lo = G->chrom_lo.data();
    G->view.n_chrom = (uint32_t)chroms.size(); G->view.n_slots = G->n_slots; G->view.d_over = 100; G->view.flags = 0;
    *out = G;
    return 0;
}
