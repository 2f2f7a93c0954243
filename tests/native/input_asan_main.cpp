// Host-sanitizer harness of the input reader (SURVEY section 5 "race detection / sanitizers": the C++ side of the step loop is the
// one piece of native HOST code with threads, mmap and raw protobuf parsing).  Built by tests/test_input_sanitizers.py with
//   hipcc --offload-host-only -fsanitize=address,undefined -g  csrc/input.hip  this file
// (csrc/common.hip holds kernels besides the error plumbing, so the two error functions input.hip needs are restated below)
// and run on CPU (no device: the reader falls back to plain host slots): opens the reader on the TFRecord files given on the command
// line, pulls batches through several passes of the data while a second thread polls las_input_records, closes the reader with
// batches still queued, and repeats for an evaluation pass and for a file with a corrupted record.  Any heap / UB finding aborts.
#include "../../include/las_hip.h"
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

static thread_local char g_err[512] = "";
void las_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap); }
extern "C" const char* las_last_error(void) { return g_err; }

static las_input_config config(int training, int rank, int world) {
    las_input_config c;
    memset(&c, 0, sizeof(c));
    const int tb[8] = {639, 1062, 1275, 1377, 1449, 1506, 1563, 1710};
    c.feat_dim = 13; c.is_training = training; c.n_bounds = 8;
    for (int i = 0; i < 8; ++i) c.bounds[i] = tb[i];
    if (!training) c.bounds[7] = 3600;
    for (int i = 0; i < 9; ++i) c.batch_limit[i] = i == 0 ? 6 : 3;      // small limits: many batches from a small corpus
    c.max_tokenlen = training ? 219 : 227; c.shuffle_buffer = training ? 3 : 0; c.cycle_length = 3;
    c.seed = 5; c.rank = rank; c.world = world; c.slots = 3;
    return c;
}

static unsigned long long checksum_u(const las_input_batch& b, int F) {
    unsigned long long s = b.B * 1000003ULL + b.T;                  // (unsigned: wrap-around is the point)
    for (int i = 0; i < b.B; ++i) {
        s = s * 31 + b.featlen[i] + b.tokenlen[i];
        const float* row = b.feat + (size_t)i * b.T * F * 3;
        s += (unsigned long long)(long long)(row[0] * 1000) + (unsigned long long)(long long)(row[(size_t)(b.featlen[i] - 1) * F * 3] * 1000);
        if (b.featlen[i] < b.T && row[(size_t)b.featlen[i] * F * 3] != 0.f) return ~0ULL;        // padding must be zero
        s += b.token[(size_t)i * b.max_tokenlen + b.tokenlen[i] - 1];
    }
    return s;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s <bad.tfrecord> <file.tfrecord>...\n", argv[0]); return 2; }
    std::vector<const char*> files(argv + 2, argv + argc);
    unsigned long long sums[2] = {0, 0};
    for (int rep = 0; rep < 2; ++rep) {                      // two readers over the same data must agree (order is a pure function of the seed)
        las_input_config c = config(1, 0, 1);
        void* r = las_input_open(files.data(), (int)files.size(), &c);
        if (!r) { fprintf(stderr, "open failed: %s\n", las_last_error()); return 1; }
        std::atomic<bool> stop{false};
        std::thread poll([&] { while (!stop) { if (las_input_records(r) < 0) break; std::this_thread::yield(); } });
        for (int k = 0; k < 60; ++k) {
            las_input_batch b;
            if (las_input_next(r, &b) != 0) { fprintf(stderr, "next failed: %s\n", las_last_error()); return 1; }
            const unsigned long long cs = checksum_u(b, 13);
            if (cs == ~0ULL) { fprintf(stderr, "non-zero padding in batch %d\n", k); return 1; }
            sums[rep] = sums[rep] * 7 + cs;
            if (las_input_release(r, b.slot) != 0) return 1;
        }
        stop = true; poll.join();
        las_input_close(r);                                 // with filled slots still queued and the producer blocked
    }
    if (sums[0] != sums[1]) { fprintf(stderr, "two readers disagree\n"); return 1; }
    for (int rank = 0; rank < 2; ++rank) {                  // lock-step shards
        las_input_config c = config(1, rank, 2);
        void* r = las_input_open(files.data(), (int)files.size(), &c);
        if (!r) return 1;
        for (int k = 0; k < 20; ++k) { las_input_batch b; if (las_input_next(r, &b) != 0) return 1; if (checksum_u(b, 13) == ~0ULL) return 1; las_input_release(r, b.slot); }
        las_input_close(r);
    }
    {   // evaluation pass: ends with rc 1
        las_input_config c = config(0, 0, 1);
        void* r = las_input_open(files.data(), (int)files.size(), &c);
        if (!r) return 1;
        int n = 0, rc;
        las_input_batch b;
        while ((rc = las_input_next(r, &b)) == 0) { ++n; las_input_release(r, b.slot); }
        if (rc != 1 || n == 0) { fprintf(stderr, "evaluation pass: rc %d after %d batches\n", rc, n); return 1; }
        las_input_close(r);
    }
    {   // corrupted framing is reported, not crashed on
        const char* bad[1] = {argv[1]};
        las_input_config c = config(0, 0, 1);
        void* r = las_input_open(bad, 1, &c);
        if (!r) return 1;
        las_input_batch b;
        int rc;
        while ((rc = las_input_next(r, &b)) == 0) las_input_release(r, b.slot);
        if (rc >= 0) { fprintf(stderr, "corruption not reported (rc %d)\n", rc); return 1; }
        las_input_close(r);
    }
    printf("INPUT_SANITIZER_OK\n");
    return 0;
}
