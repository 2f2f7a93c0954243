// input.hip -- the input side of the train / eval step loop: TFRecord reader + length bucketing + pinned batch ring + H2D upload.
//
// Host code (no kernels).  Counterpart of the reference's tf.data pipeline, tfrecord_data_loader.py:54-109
// (list_files -> parallel_interleave(cycle_length 16) -> map(data_parser, 16 parallel calls) ->
// bucket_by_sequence_length(pad_to_bucket_boundary) -> shuffle(64) -> repeat -> prefetch): what the reference gets from TensorFlow's
// C++ runtime threads, one producer thread per reader does here -- the files are mmap'ed, a record's float payload is only touched
// when the batch that contains it is assembled (one memcpy into a pinned slot), so a single thread sustains several GB/s, well above
// the ~0.5 GB/s a 3,200 utterances/s train loop consumes.  The batch ORDER is a pure function of (files, seed): the interleave is
// the reference's deterministic round robin (block_length 1), the shuffles use the splitmix64 stream that
// tfrecord_data_loader._Rng restates in Python, so the Python iterator and this reader produce identical batches (tested).
//
// Data parallelism (north_star "shards the TFRecord utterance batch across the GPUs"): with world > 1 every rank
// scans the SAME record stream, a bucket emits when it holds world x batch_limit utterances, and rank r keeps rows r, r + world, ...
// of that global batch.  All ranks therefore run the same bucket (same T) at every step -- no rank waits for another rank's longer
// utterances in the gradient all-reduce -- and a rank never touches the payload pages of the rows it does not keep.
#include "las_common.h"
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <mutex>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

namespace {

// ---- crc32c (Castagnoli), masked as the TFRecord framing does.  The 8-byte length is verified when a record is framed, the
// payload when the batch that contains it is assembled (only the rows this rank keeps: a rank never touches the other rows' pages).
// Eight bytes per step: the host's crc32 instruction when it has one, slicing-by-8 tables otherwise.
uint32_t crc_tab[8][256];
bool crc_hw = false;
void crc_init() {
    static std::once_flag once;
    std::call_once(once, [] {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            crc_tab[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t) crc_tab[t][i] = crc_tab[0][crc_tab[t - 1][i] & 0xFF] ^ (crc_tab[t - 1][i] >> 8);
#if defined(__x86_64__)
        crc_hw = __builtin_cpu_supports("sse4.2");
#endif
    });
}
#if defined(__x86_64__)
__attribute__((target("sse4.2"))) uint32_t crc_update_hw(uint32_t c, const uint8_t* p, size_t n) {
    uint64_t c64 = c;
    while (n && ((uintptr_t)p & 7)) { c64 = __builtin_ia32_crc32qi((uint32_t)c64, *p++); --n; }
    for (; n >= 8; n -= 8, p += 8) { uint64_t w; memcpy(&w, p, 8); c64 = __builtin_ia32_crc32di(c64, w); }
    while (n--) c64 = __builtin_ia32_crc32qi((uint32_t)c64, *p++);
    return (uint32_t)c64;
}
#endif
uint32_t crc_update(uint32_t c, const uint8_t* p, size_t n) {
#if defined(__x86_64__)
    if (crc_hw) return crc_update_hw(c, p, n);
#endif
    for (; n >= 8; n -= 8, p += 8) {
        uint32_t lo, hi; memcpy(&lo, p, 4); memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = crc_tab[7][lo & 0xFF] ^ crc_tab[6][(lo >> 8) & 0xFF] ^ crc_tab[5][(lo >> 16) & 0xFF] ^ crc_tab[4][lo >> 24] ^
            crc_tab[3][hi & 0xFF] ^ crc_tab[2][(hi >> 8) & 0xFF] ^ crc_tab[1][(hi >> 16) & 0xFF] ^ crc_tab[0][hi >> 24];
    }
    while (n--) c = crc_tab[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return c;
}
uint32_t masked_crc32c(const uint8_t* p, size_t n) {
    const uint32_t c = crc_update(0xFFFFFFFFu, p, n) ^ 0xFFFFFFFFu;
    return ((c >> 15) | (c << 17)) + 0xA282EAD8u;
}

// ---- the shared pseudo-random stream (tfrecord_data_loader._Rng): splitmix64 ----------------------------------------------------
struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    uint64_t below(uint64_t n) { return next() % n; }
    template <class T> void shuffle(std::vector<T>& v) {                 // Fisher-Yates from the top
        for (size_t i = v.size(); i > 1; --i) std::swap(v[i - 1], v[below(i)]);
    }
};

// ---- protobuf: just enough of tf.train.Example ------------------------------------------------------------------------------------
struct Cursor {
    const uint8_t* p; const uint8_t* e; bool ok = true;
    uint64_t varint() {
        uint64_t v = 0; int sh = 0;
        while (p < e) { uint8_t b = *p++; v |= (uint64_t)(b & 0x7F) << sh; if (!(b & 0x80)) return v; sh += 7; if (sh > 63) break; }
        ok = false; return 0;
    }
    // next field: number, wire type, and for length-delimited fields the sub-range
    bool field(int& fn, int& wt, Cursor& sub, uint64_t& val) {
        if (p >= e || !ok) return false;
        uint64_t key = varint(); fn = (int)(key >> 3); wt = (int)(key & 7);
        if (wt == 0) val = varint();
        else if (wt == 2) { uint64_t n = varint(); if (!ok || n > (uint64_t)(e - p)) { ok = false; return false; } sub = {p, p + n}; p += n; }
        else if (wt == 5) { if (e - p < 4) { ok = false; return false; } sub = {p, p + 4}; p += 4; }
        else if (wt == 1) { if (e - p < 8) { ok = false; return false; } sub = {p, p + 8}; p += 8; }
        else { ok = false; return false; }
        return ok;
    }
};

struct Record {              // one parsed utterance: pointers into the mmap'ed file, nothing copied yet
    const uint8_t* feat = nullptr; size_t nfeat = 0;   // packed little-endian floats at ANY byte offset of the file: kept as bytes, memcpy only
    int T = 0, F = 0;
    std::vector<int> token;
    const uint8_t* payload = nullptr; size_t npayload = 0;   // the framed record (its trailing masked crc32c sits right behind it)
};

bool parse_int64_list(Cursor c, std::vector<long long>& out) {
    int fn, wt; Cursor sub{nullptr, nullptr}; uint64_t val;
    while (c.field(fn, wt, sub, val)) {
        if (fn != 1) continue;
        if (wt == 2) { while (sub.p < sub.e && sub.ok) out.push_back((long long)sub.varint()); if (!sub.ok) return false; }
        else if (wt == 0) out.push_back((long long)val);
    }
    return c.ok;
}

bool parse_example(const uint8_t* p, size_t n, Record& r) {
    Cursor ex{p, p + n};
    int fn, wt; Cursor feats{nullptr, nullptr}; uint64_t val;
    std::vector<long long> shape, token;
    while (ex.field(fn, wt, feats, val)) {
        if (fn != 1 || wt != 2) continue;                                   // Example.features
        int f2, w2; Cursor entry{nullptr, nullptr};
        while (feats.field(f2, w2, entry, val)) {
            if (f2 != 1 || w2 != 2) continue;                               // map<string, Feature> entry
            int f3, w3; Cursor v3{nullptr, nullptr};
            std::string key; Cursor feature{nullptr, nullptr}; bool have = false;
            while (entry.field(f3, w3, v3, val)) {
                if (f3 == 1 && w3 == 2) key.assign((const char*)v3.p, v3.e - v3.p);
                else if (f3 == 2 && w3 == 2) { feature = v3; have = true; }
            }
            if (!entry.ok) return false;
            if (!have) continue;
            int f4, w4; Cursor lst{nullptr, nullptr};
            while (feature.field(f4, w4, lst, val)) {
                if (w4 != 2) continue;
                if (f4 == 2 && key == "feat") {                             // FloatList: one packed run (what TF writes)
                    int f5, w5; Cursor run{nullptr, nullptr};
                    while (lst.field(f5, w5, run, val))
                        if (f5 == 1 && w5 == 2) { r.feat = run.p; r.nfeat = (size_t)(run.e - run.p) / 4; }
                    if (!lst.ok) return false;
                } else if (f4 == 3 && key == "shape") { if (!parse_int64_list(lst, shape)) return false; }
                else if (f4 == 3 && key == "token") { if (!parse_int64_list(lst, token)) return false; }
            }
            if (!feature.ok) return false;
        }
        if (!feats.ok) return false;
    }
    if (!ex.ok || shape.size() != 3 || !r.feat) return false;
    r.T = (int)shape[0]; r.F = (int)shape[1];
    if (shape[2] != 3 || (size_t)r.T * r.F * 3 != r.nfeat) return false;
    r.token.assign(token.begin(), token.end());
    return true;
}

// ---- files ---------------------------------------------------------------------------------------------------------------------------
struct MappedFile {
    const uint8_t* base = nullptr; size_t size = 0, pos = 0; std::string path;
    bool open(const std::string& p_) {
        path = p_;
        int fd = ::open(p_.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) { ::close(fd); return false; }
        size = (size_t)st.st_size;
        if (size) {
            void* m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { ::close(fd); return false; }
            base = (const uint8_t*)m;
            madvise(m, size, MADV_SEQUENTIAL);
        }
        ::close(fd);
        return true;
    }
    void close() { if (base) munmap((void*)base, size); base = nullptr; }
    // 1: a record, 0: end of file, -1: corrupt
    int next(const uint8_t*& payload, size_t& n) {
        if (pos == size) return 0;
        if (size - pos < 16) return -1;                                    // header + both checksums do not fit: a truncated file
        uint64_t len; uint32_t lcrc;
        memcpy(&len, base + pos, 8); memcpy(&lcrc, base + pos + 8, 4);
        if (masked_crc32c(base + pos, 8) != lcrc) return -1;
        if (len > size - pos - 16) return -1;                              // (size - pos >= 16 was checked: no wrap-around)
        payload = base + pos + 12; n = (size_t)len;
        pos += 12 + len + 4;
        return 1;
    }
};

struct Slot {
    float* feat = nullptr; int* token = nullptr;       // pinned (hipHostMalloc) when a device is present, else plain host memory
    std::vector<int> featlen, tokenlen;
    int B = 0, T = 0, bucket = 0, global_B = 0;
    hipEvent_t ev = nullptr; bool ev_pending = false;
    int state = 0;                                     // 0 free, 1 filled (waiting for las_input_next), 2 handed out
};

struct Pending { std::vector<Record> items; int bucket; };       // a complete (global) batch waiting in the shuffle buffer

struct Reader {
    las_input_config cfg;
    std::vector<std::string> files;
    std::vector<Slot> slots;
    size_t feat_cap = 0;
    bool pinned = false;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv_filled, cv_free;
    std::vector<int> order;                            // filled slots in production order
    bool stop = false, finished = false;
    std::string error;
    long long batches = 0;
    std::atomic<long long> records{0};

    int limit(int k) const { return cfg.batch_limit[k] * std::max(cfg.world, 1); }

    bool fail(const std::string& m) {
        std::lock_guard<std::mutex> l(mu);
        error = m; finished = true;
        cv_filled.notify_all();
        return false;
    }

    // fills one slot with this rank's rows of a global batch; blocks while no slot is free.  false: stop requested
    bool emit(Pending& pb) {
        const int world = std::max(cfg.world, 1), rank = cfg.rank;
        std::vector<const Record*> mine;
        for (size_t i = rank; i < pb.items.size(); i += world) mine.push_back(&pb.items[i]);
        if (pb.items.size() < (size_t)world) return true;                 // a leftover smaller than the world: dropped on EVERY rank
        for (const Record* r : mine) {                                    // payload checksum of the rows this rank keeps
            uint32_t want; memcpy(&want, r->payload + r->npayload, 4);
            if (masked_crc32c(r->payload, r->npayload) != want) return fail("corrupted TFRecord payload (crc32c mismatch)");
        }
        const int k = pb.bucket;
        int T = cfg.bounds[k] - 1;
        int s = -1;
        {
            std::unique_lock<std::mutex> l(mu);
            cv_free.wait(l, [&] { if (stop) return true; for (size_t i = 0; i < slots.size(); ++i) if (slots[i].state == 0) { s = (int)i; return true; } return false; });
            if (stop) return false;
            slots[s].state = 3;                                           // being filled
        }
        Slot& sl = slots[s];
        if (sl.ev_pending) { (void)hipEventSynchronize(sl.ev); sl.ev_pending = false; }      // the previous upload out of this slot
        const int F = cfg.feat_dim, B = (int)mine.size();
        const size_t row = (size_t)T * F * 3;
        sl.B = B; sl.T = T; sl.bucket = k; sl.global_B = (int)pb.items.size();
        sl.featlen.assign(B, 0); sl.tokenlen.assign(B, 0);
        memset(sl.token, 0, sizeof(int) * (size_t)B * cfg.max_tokenlen);
        for (int b = 0; b < B; ++b) {
            const Record& r = *mine[b];
            const size_t n = (size_t)r.T * F * 3;
            memcpy(sl.feat + b * row, r.feat, n * 4);
            memset(sl.feat + b * row + n, 0, (row - n) * 4);
            sl.featlen[b] = r.T;
            sl.tokenlen[b] = (int)r.token.size();
            memcpy(sl.token + (size_t)b * cfg.max_tokenlen, r.token.data(), sizeof(int) * r.token.size());
        }
        {
            std::lock_guard<std::mutex> l(mu);
            sl.state = 1;
            order.push_back(s);
            ++batches;
        }
        cv_filled.notify_all();
        return true;
    }

    void run() {
        crc_init();
        Rng rng(cfg.seed);
        const int nb = cfg.n_bounds;
        std::vector<Pending> shuf;
        do {
            // ---- one pass over the data
            std::vector<std::string> fl = files;
            rng.shuffle(fl);
            std::vector<MappedFile*> active;
            size_t nextf = 0;
            std::vector<std::vector<Record>> buckets(nb);
            auto out = [&](Pending&& pb) -> bool {
                if (cfg.shuffle_buffer > 0) {
                    shuf.push_back(std::move(pb));
                    if ((int)shuf.size() > cfg.shuffle_buffer) {
                        const size_t j = rng.below(shuf.size());
                        Pending q = std::move(shuf[j]);
                        shuf.erase(shuf.begin() + j);
                        return emit(q);
                    }
                    return true;
                }
                return emit(pb);
            };
            bool ok = true;
            for (;;) {
                while ((int)active.size() < cfg.cycle_length && nextf < fl.size()) {
                    MappedFile* mf = new MappedFile();
                    if (!mf->open(fl[nextf])) { delete mf; fail("cannot open " + fl[nextf]); ok = false; break; }
                    ++nextf;
                    active.push_back(mf);
                }
                if (!ok || active.empty()) break;
                std::vector<MappedFile*> round = active;                   // block_length 1 round robin
                for (MappedFile* mf : round) {
                    const uint8_t* p; size_t n;
                    const int rc = mf->next(p, n);
                    if (rc == 0) { active.erase(std::find(active.begin(), active.end(), mf)); retired.push_back(mf); continue; }
                    if (rc < 0) { fail("corrupted TFRecord framing in " + mf->path); ok = false; break; }
                    Record r;
                    if (!parse_example(p, n, r) || r.F != cfg.feat_dim) { fail("malformed Example in " + mf->path); ok = false; break; }
                    r.payload = p; r.npayload = n;
                    ++records;
                    const int k = (int)(std::upper_bound(cfg.bounds, cfg.bounds + nb, r.T) - cfg.bounds);
                    if (k >= nb) { fail("utterance of " + std::to_string(r.T) + " frames exceeds the last bucket boundary " + std::to_string(cfg.bounds[nb - 1])); ok = false; break; }
                    if ((int)r.token.size() > cfg.max_tokenlen) { fail("token sequence of " + std::to_string(r.token.size()) + " exceeds the padded length " + std::to_string(cfg.max_tokenlen)); ok = false; break; }
                    buckets[k].push_back(std::move(r));
                    if ((int)buckets[k].size() == limit(k)) {
                        Pending pb{std::move(buckets[k]), k};
                        buckets[k].clear();
                        if (!out(std::move(pb))) { ok = false; break; }
                    }
                }
                if (!ok) break;
            }
            if (ok)
                for (int k = 0; k < nb && ok; ++k)                          // leftovers at the end of the data
                    if (!buckets[k].empty()) { Pending pb{std::move(buckets[k]), k}; ok = out(std::move(pb)); }
            if (ok)
                while (!shuf.empty() && ok) {
                    const size_t j = rng.below(shuf.size());
                    Pending q = std::move(shuf[j]);
                    shuf.erase(shuf.begin() + j);
                    ok = emit(q);
                }
            for (MappedFile* mf : active) { mf->close(); delete mf; }
            // (the maps of finished files stay until here: pending batches point into them)
            for (MappedFile* mf : retired) { mf->close(); delete mf; }
            retired.clear();
            if (!ok) break;
            if (records == 0) { fail("no records in the input files"); break; }
        } while (cfg.is_training && !stop);
        std::lock_guard<std::mutex> l(mu);
        finished = true;
        cv_filled.notify_all();
    }
    std::vector<MappedFile*> retired;
};

}  // namespace

extern "C" unsigned int las_crc32c(const void* data, size_t n) {
    crc_init();
    return crc_update(0xFFFFFFFFu, (const uint8_t*)data, n) ^ 0xFFFFFFFFu;
}

static void destroy(Reader* r) {                       // slots (whatever part of them exists) + the reader itself
    for (Slot& s : r->slots) {
        if (s.ev_pending) (void)hipEventSynchronize(s.ev);
        if (r->pinned) { if (s.feat) (void)hipHostFree(s.feat); if (s.token) (void)hipHostFree(s.token); if (s.ev) (void)hipEventDestroy(s.ev); }
        else { free(s.feat); free(s.token); }
    }
    delete r;
}

extern "C" void* las_input_open(const char* const* files, int nfiles, const las_input_config* cfg) {
    if (!files || nfiles <= 0 || !cfg) { las_set_error("las_input_open: no files / no configuration"); return nullptr; }
    if (cfg->feat_dim <= 0 || cfg->n_bounds <= 0 || cfg->n_bounds > 16 || cfg->max_tokenlen <= 0 || cfg->cycle_length <= 0 ||
        cfg->world < 1 || cfg->rank < 0 || cfg->rank >= cfg->world || cfg->slots < 2 || cfg->slots > 64) {
        las_set_error("las_input_open: bad configuration"); return nullptr;
    }
    for (int k = 0; k < cfg->n_bounds; ++k)
        if (cfg->batch_limit[k] <= 0 || cfg->bounds[k] <= 1 || (k && cfg->bounds[k] <= cfg->bounds[k - 1])) {
            las_set_error("las_input_open: bucket boundaries must increase and batch limits be positive"); return nullptr;
        }
    if (!cfg->is_training && cfg->world > 1) {       // an end-of-data leftover smaller than the world would be skipped silently
        las_set_error("las_input_open: evaluation data is not sharded by the reader (is_training = 0 needs world = 1): read every "
                      "batch and take this rank's share (las.parallel.shard; test.py / decode.py do)");
        return nullptr;
    }
    Reader* r = new Reader();
    r->cfg = *cfg;
    for (int i = 0; i < nfiles; ++i) r->files.emplace_back(files[i]);
    size_t cap = 0; int maxB = 0;
    for (int k = 0; k < cfg->n_bounds; ++k) {
        cap = std::max(cap, (size_t)cfg->batch_limit[k] * (cfg->bounds[k] - 1) * cfg->feat_dim * 3);
        maxB = std::max(maxB, cfg->batch_limit[k]);
    }
    r->feat_cap = cap;
    r->slots.resize(cfg->slots);
    int ndev = 0;
    r->pinned = hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0;
    for (Slot& s : r->slots) {
        const size_t fb = cap * sizeof(float), tb = (size_t)maxB * cfg->max_tokenlen * sizeof(int);
        if (r->pinned && (hipHostMalloc((void**)&s.feat, fb, hipHostMallocDefault) != hipSuccess ||
                          hipHostMalloc((void**)&s.token, tb, hipHostMallocDefault) != hipSuccess ||
                          hipEventCreateWithFlags(&s.ev, hipEventDisableTiming) != hipSuccess)) {
            las_set_error("las_input_open: pinned host allocation failed"); destroy(r); return nullptr;
        }
        if (!r->pinned) { s.feat = (float*)aligned_alloc(4096, (fb + 4095) / 4096 * 4096); s.token = (int*)aligned_alloc(4096, (tb + 4095) / 4096 * 4096); }
        if (!s.feat || !s.token) { las_set_error("las_input_open: host allocation failed"); destroy(r); return nullptr; }
    }
    (void)hipGetLastError();
    r->th = std::thread([r] { r->run(); });
    return r;
}

extern "C" int las_input_next(void* h, las_input_batch* out) {
    LAS_ARG(h && out, "las_input_next: null argument");
    Reader* r = (Reader*)h;
    std::unique_lock<std::mutex> l(r->mu);
    r->cv_filled.wait(l, [&] { return !r->order.empty() || r->finished; });
    if (r->order.empty()) {
        if (!r->error.empty()) { las_set_error("las_input: %s", r->error.c_str()); return -1; }
        return 1;                                                           // end of the (evaluation) data
    }
    const int s = r->order.front();
    r->order.erase(r->order.begin());
    Slot& sl = r->slots[s];
    sl.state = 2;
    out->slot = s; out->B = sl.B; out->T = sl.T; out->bucket = sl.bucket; out->global_B = sl.global_B;
    out->max_tokenlen = r->cfg.max_tokenlen;
    out->feat = sl.feat; out->token = sl.token; out->featlen = sl.featlen.data(); out->tokenlen = sl.tokenlen.data();
    return 0;
}

extern "C" int las_input_upload(void* h, int slot, float* d_feat, int* d_token, void* stream) {
    LAS_ARG(h && d_feat && d_token, "las_input_upload: null argument");
    Reader* r = (Reader*)h;
    LAS_ARG(slot >= 0 && slot < (int)r->slots.size() && r->slots[slot].state == 2, "las_input_upload: slot %d was not handed out", slot);
    LAS_ARG(r->pinned, "las_input_upload: no ROCm device (the reader was opened without one)");
    Slot& sl = r->slots[slot];
    LAS_HIP(hipMemcpyAsync(d_feat, sl.feat, sizeof(float) * (size_t)sl.B * sl.T * r->cfg.feat_dim * 3, hipMemcpyHostToDevice, (hipStream_t)stream));
    LAS_HIP(hipMemcpyAsync(d_token, sl.token, sizeof(int) * (size_t)sl.B * r->cfg.max_tokenlen, hipMemcpyHostToDevice, (hipStream_t)stream));
    LAS_HIP(hipEventRecord(sl.ev, (hipStream_t)stream));
    sl.ev_pending = true;                                                   // the producer waits for it before it refills the slot
    return 0;
}

extern "C" int las_input_release(void* h, int slot) {
    LAS_ARG(h, "las_input_release: null handle");
    Reader* r = (Reader*)h;
    LAS_ARG(slot >= 0 && slot < (int)r->slots.size(), "las_input_release: bad slot");
    {
        std::lock_guard<std::mutex> l(r->mu);
        LAS_ARG(r->slots[slot].state == 2, "las_input_release: slot %d was not handed out", slot);
        r->slots[slot].state = 0;
    }
    r->cv_free.notify_all();
    return 0;
}

extern "C" long long las_input_records(void* h) {
    if (!h) return -1;
    return ((Reader*)h)->records.load();
}

extern "C" void las_input_close(void* h) {
    if (!h) return;
    Reader* r = (Reader*)h;
    {
        std::lock_guard<std::mutex> l(r->mu);
        r->stop = true;
    }
    r->cv_free.notify_all();
    if (r->th.joinable()) r->th.join();
    destroy(r);
}
