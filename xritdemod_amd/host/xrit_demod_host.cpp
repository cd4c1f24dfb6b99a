// xrit_demod_host -- the reference demodulator's plumbing around libxritdemod_amd:
// IQ file source -> demodulation chain on the GPU -> int8 soft symbols -> TCP client
// socket towards a decoder (or a file).  It is the host loop SURVEY.md section 8(b)/(f)
// asks for, so that the unmodified xritDecoder listening on :5000 can attach.
//
// What it mirrors of /root/reference/demodulator/src:
//   * CFileFrontend.cpp:35-60   blocks of complex samples read from a raw IQ file; with
//                               --paced the blocks are released at the capture's sample
//                               rate like the reference does (default: as fast as possible)
//   * demodulator.cpp:100-168   processSamples(): one chain call per block, state carried
//   * demodulator.cpp:38,       --fifo: the reference's chunking rule instead of fixed --block calls: the source delivers blocks of
//     :108-119, Parameters.h:57 --fifo-block samples (65535: CFileFrontend's BUFFERSIZE) into a FIFO of 1 Mi floats (512 Ki complex
//                               samples); the DSP loop looks at it after every --fifo-lag blocks and, once it holds 64 Ki floats,
//                               takes EVERYTHING it holds as one chunk -- so the chunk length varies, and with it which `length
//                               mod decimation` samples :137 drops.  What does not fit the FIFO is lost ("Input Samples Fifo is
//                               overflowing!").  The reference's chunk sizes depend on thread timing; here they are a function of
//                               (--fifo-block, --fifo-lag), so that a run can be repeated and checked against the oracle.
//   * SymbolManager.cpp:37-52   soft symbol -> int8 (x127, clamp, C-cast truncation), sent
//                               in pieces of at most 16384 bytes (SM_SOCKET_BUFFER_SIZE)
//   * SymbolManager.cpp:23-35   the demodulator is the TCP *client*; it retries the
//                               connection once per second until the decoder listens
//   * SymbolManager.cpp:78-106  --drop: the reference's queue between the DSP thread and the sender thread, with its
//                               loss behaviour: a chunk of symbols is DROPPED when 1 Mi symbols are already queued
//                               ("SymbolManager Buffer is full!!! Dropping samples."), and everything queued is
//                               dropped while no decoder is connected.  The default (no --drop) is lossless: the DSP
//                               loop waits for the socket, which is what a file run at GPU speed wants.
//   * decoder/src/Statistics.h:34  --stats reports the queue's fill as demodulatorFifoUsage (percent of its capacity,
//                               one byte like the decoder's statistics field) with its peak and the drop counts
//   * DiagManager.cpp:24-62,    --diag: the constellation tap -- after every chain call up to 1024 floats of
//     demodulator.cpp:161-163   the complex symbols (I, Q interleaved) are queued; whenever 1024 are queued and
//                               10 ms have passed they go out as int8 (x128, clamp, truncation) in one UDP
//                               datagram from port 9001 to HOST:9000
// Only the C ABI of include/xritdemod_amd.h is used (no HIP headers): this file builds
// with plain g++.
#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <deque>
#include <mutex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/xritdemod_amd.h"

namespace {

struct Options {
    std::string mode = "lrit";
    std::string input;
    std::string format = "cf32";
    std::string sink = "tcp://127.0.0.1:5000";
    std::string diag;              // udp://HOST:PORT, empty = off
    double sample_rate = 2.5e6;
    unsigned decimation = 1;
    size_t block = 1u << 22;       // complex samples per chain call (>= 4 Mi when throughput matters)
    int device = 0;
    int connect_tries = 30;
    bool paced = false;
    bool stats = false;
    bool drop = false;             // SymbolManager's queue and loss behaviour instead of back-pressure
    size_t queue_symbols = 1024 * 1024;   // SM_MAX_SYMBOL_BUFFER (SymbolManager.h:23)
    int sndbuf = 0;                // > 0: SO_SNDBUF of the decoder socket (the kernel's default grows to megabytes)
    int gpus = 1;                  // > 1: every block is cut in that many time slices, one per GPU (xrit_group_*)
    bool fifo = false;             // the reference's FIFO chunking (demodulator.cpp:108-119) instead of fixed --block calls
    size_t fifo_block = 65535;     // complex samples per source callback (CFileFrontend.cpp:12 BUFFERSIZE)
    int fifo_lag = 1;              // source blocks that arrive between two looks of the DSP loop
    int front_exact = 0;           // cfg.front_exact: 1 = the Costas loop's final pass warmed up; 2 = the front end bit for bit a CPU chain's
};
constexpr size_t FIFO_COMPLEX = 1024 * 1024 / 2;      // FIFO_SIZE floats (Parameters.h:57)
constexpr size_t FIFO_MIN_COMPLEX = 64 * 1024 / 2;    // "Lets wait for more samples" (demodulator.cpp:113)

void usage()
{
    std::fprintf(stderr,
                 "usage: xrit_demod_host --input FILE [--format cf32|s16|s8|u8] [--mode lrit|hrit]\n"
                 "         [--sample-rate HZ] [--decimation D] [--block SAMPLES] [--device N]\n"
                 "         [--sink tcp://HOST:PORT | file:PATH | null] [--connect-tries N] [--paced] [--stats]\n"
                 "         [--diag udp://HOST:PORT] [--drop [--queue-symbols N]] [--sndbuf BYTES]\n"
                 "         [--gpus N]   (devices 0..N-1: each block of --block samples is cut in N time slices, RCCL edge exchange)\n"
                 "         [--fifo [--fifo-block SAMPLES] [--fifo-lag BLOCKS]]   (the reference's FIFO chunking, demodulator.cpp:108-119)\n"
                 "         [--front-exact [-1|1|2]]   (cfg.front_exact; default 0: the bit-exact front end on blocks of less than a million symbols;\n"
                 "                                  -1: the fast one always; 1: the Costas loop's final pass warmed up, ~12 %% slower on big blocks;\n"
                 "                                  2: filters, AGC and Costas loop bit for bit a CPU chain's -- soft symbols within 1e-4 rms of it\n"
                 "                                  on every configuration, ~3 x slower on big blocks)\n");
}

bool parse(int argc, char **argv, Options &o)
{
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto need = [&](const char *name) -> const char * {
            if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", name); return nullptr; }
            return argv[++i];
        };
        const char *v = nullptr;
        if (a == "--input") { if (!(v = need("--input"))) return false; o.input = v; }
        else if (a == "--format") { if (!(v = need("--format"))) return false; o.format = v; }
        else if (a == "--mode") { if (!(v = need("--mode"))) return false; o.mode = v; }
        else if (a == "--sink") { if (!(v = need("--sink"))) return false; o.sink = v; }
        else if (a == "--sample-rate") { if (!(v = need("--sample-rate"))) return false; o.sample_rate = std::atof(v); }
        else if (a == "--decimation") { if (!(v = need("--decimation"))) return false; o.decimation = (unsigned)std::atoi(v); }
        else if (a == "--block") { if (!(v = need("--block"))) return false; o.block = (size_t)std::atoll(v); }
        else if (a == "--device") { if (!(v = need("--device"))) return false; o.device = std::atoi(v); }
        else if (a == "--connect-tries") { if (!(v = need("--connect-tries"))) return false; o.connect_tries = std::atoi(v); }
        else if (a == "--diag") { if (!(v = need("--diag"))) return false; o.diag = v; }
        else if (a == "--queue-symbols") { if (!(v = need("--queue-symbols"))) return false; o.queue_symbols = (size_t)std::atoll(v); }
        else if (a == "--sndbuf") { if (!(v = need("--sndbuf"))) return false; o.sndbuf = std::atoi(v); }
        else if (a == "--gpus") { if (!(v = need("--gpus"))) return false; o.gpus = std::atoi(v); }
        else if (a == "--fifo-block") { if (!(v = need("--fifo-block"))) return false; o.fifo_block = (size_t)std::atoll(v); }
        else if (a == "--fifo-lag") { if (!(v = need("--fifo-lag"))) return false; o.fifo_lag = std::atoi(v); }
        else if (a == "--fifo") o.fifo = true;
        else if (a == "--front-exact") {
            o.front_exact = 1;
            if (i + 1 < argc && (std::string(argv[i + 1]) == "1" || std::string(argv[i + 1]) == "2" || std::string(argv[i + 1]) == "-1")) o.front_exact = atoi(argv[++i]);
        }
        else if (a == "--drop") o.drop = true;
        else if (a == "--paced") o.paced = true;
        else if (a == "--stats") o.stats = true;
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return false; }
    }
    if (o.fifo && (o.fifo_block < 1 || o.fifo_block > FIFO_COMPLEX || o.fifo_lag < 1 || o.gpus > 1)) {
        std::fprintf(stderr, "--fifo: --fifo-block 1..%zu, --fifo-lag >= 1, one GPU\n", FIFO_COMPLEX);
        return false;
    }
    return !o.input.empty() && o.block > 0 && o.decimation >= 1 && o.gpus >= 1;
}

// ---- sink ------------------------------------------------------------------------------------------------
struct Sink {
    enum Kind { NONE, TCP, FILE_ } kind = NONE;
    int fd = -1;
    FILE *f = nullptr;
    std::string host;
    int port = 0;
    int tries = 30;
    int sndbuf = 0;
    size_t sent = 0;

    bool open(const std::string &spec, int connect_tries)
    {
        tries = connect_tries;
        if (spec == "null") { kind = NONE; return true; }
        if (spec.rfind("file:", 0) == 0) {
            kind = FILE_;
            f = std::fopen(spec.c_str() + 5, "wb");
            if (!f) { std::perror("sink file"); return false; }
            return true;
        }
        if (spec.rfind("tcp://", 0) == 0) {
            kind = TCP;
            std::string hp = spec.substr(6);
            size_t c = hp.rfind(':');
            if (c == std::string::npos) { std::fprintf(stderr, "sink: tcp://HOST:PORT expected\n"); return false; }
            host = hp.substr(0, c);
            port = std::atoi(hp.c_str() + c + 1);
            return connect_retry();
        }
        std::fprintf(stderr, "sink: unknown spec %s\n", spec.c_str());
        return false;
    }
    bool connect_once()
    {
        addrinfo hints{}, *res = nullptr;
        hints.ai_family = AF_INET;
        hints.ai_socktype = SOCK_STREAM;
        if (getaddrinfo(host.c_str(), std::to_string(port).c_str(), &hints, &res) != 0 || !res) return false;
        fd = ::socket(res->ai_family, res->ai_socktype, res->ai_protocol);
        if (fd >= 0 && sndbuf > 0) (void)setsockopt(fd, SOL_SOCKET, SO_SNDBUF, &sndbuf, sizeof sndbuf);
        bool ok = fd >= 0 && ::connect(fd, res->ai_addr, res->ai_addrlen) == 0;
        freeaddrinfo(res);
        if (!ok) {
            if (fd >= 0) ::close(fd);
            fd = -1;
            return false;
        }
        int one = 1;
        (void)setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
        return true;
    }
    // SymbolManager.cpp:23-35: try, sleep one second, try again
    bool connect_retry()
    {
        for (int t = 0; t < tries; ++t) {
            std::fprintf(stderr, "Trying to connect to decoder at %s:%d\n", host.c_str(), port);
            if (connect_once()) return true;
            std::this_thread::sleep_for(std::chrono::seconds(1));
        }
        std::fprintf(stderr, "sink: no decoder at %s:%d\n", host.c_str(), port);
        return false;
    }
    bool send_all(const int8_t *p, size_t n)
    {
        sent += n;
        if (kind == NONE) return true;
        if (kind == FILE_) return std::fwrite(p, 1, n, f) == n;
        // pieces of at most SM_SOCKET_BUFFER_SIZE bytes, like SymbolManager::process
        while (n > 0) {
            size_t piece = n > 16384 ? 16384 : n;
            ssize_t w = ::send(fd, p, piece, MSG_NOSIGNAL);
            if (w <= 0) {
                std::fprintf(stderr, "Disconnected from decoder.\n");
                ::close(fd);
                fd = -1;
                if (!connect_retry()) return false;
                continue;
            }
            p += w;
            n -= (size_t)w;
        }
        return true;
    }
    void close_all()
    {
        if (fd >= 0) ::close(fd);
        if (f) std::fclose(f);
        fd = -1;
        f = nullptr;
    }
};

// ---- SymbolManager's queue (--drop) ---------------------------------------------------------------------------
// SymbolManager::add / ::process (SymbolManager.cpp:37-52,78-106): the DSP thread queues soft symbols, the sender
// thread takes at most 16384 of them at a time, quantises (x127, clamp, C cast) and sends.  Two ways to lose
// symbols, both kept: add() drops its whole chunk when the queue already holds `cap` symbols, process() empties
// the queue while no decoder is connected.
struct SymbolQueue {
    std::deque<float> q;
    std::mutex m;
    size_t cap = 1024 * 1024;
    size_t dropped_full = 0, dropped_disconnected = 0, peak = 0;

    void add(const float *sym, size_t n)
    {
        std::lock_guard<std::mutex> g(m);
        if (q.size() >= cap) {
            std::fprintf(stderr, "SymbolManager Buffer is full!!! Dropping samples.\n");
            dropped_full += n;
            return;
        }
        q.insert(q.end(), sym, sym + n);
        if (q.size() > peak) peak = q.size();
    }
    size_t take(int8_t *out, size_t max)
    {
        std::lock_guard<std::mutex> g(m);
        const size_t n = q.size() < max ? q.size() : max;
        for (size_t i = 0; i < n; ++i) {
            float f = q.front() * 127;
            f = f > 127 ? 127 : f;
            f = f < -128 ? -128 : f;
            out[i] = (int8_t)(char)f;
            q.pop_front();
        }
        return n;
    }
    void clear_disconnected()
    {
        std::lock_guard<std::mutex> g(m);
        dropped_disconnected += q.size();
        q.clear();
    }
    size_t size()
    {
        std::lock_guard<std::mutex> g(m);
        return q.size();
    }
    // decoder/src/Statistics.h:34: one byte, percent of the capacity (a chunk may overshoot the capacity: clamped)
    static unsigned usage(size_t fill, size_t cap)
    {
        const size_t pc = cap ? fill * 100 / cap : 0;
        return (unsigned)(pc > 255 ? 255 : pc);
    }
};

// ---- constellation tap (DiagManager) -----------------------------------------------------------------------
struct Diag {
    int fd = -1;
    sockaddr_in to{};
    std::deque<float> q;
    std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now() - std::chrono::seconds(1);
    static constexpr size_t CAP = 1u << 16;            // the reference's buffer simply stops taking samples when full

    bool open(const std::string &spec)
    {
        if (spec.rfind("udp://", 0) != 0) { std::fprintf(stderr, "diag: udp://HOST:PORT expected\n"); return false; }
        std::string hp = spec.substr(6);
        size_t c = hp.rfind(':');
        if (c == std::string::npos) return false;
        addrinfo hints{}, *res = nullptr;
        hints.ai_family = AF_INET;
        hints.ai_socktype = SOCK_DGRAM;
        if (getaddrinfo(hp.substr(0, c).c_str(), hp.c_str() + c + 1, &hints, &res) != 0 || !res) return false;
        std::memcpy(&to, res->ai_addr, sizeof to);
        freeaddrinfo(res);
        fd = ::socket(AF_INET, SOCK_DGRAM, 0);
        if (fd < 0) return false;
        sockaddr_in from{};
        from.sin_family = AF_INET;
        from.sin_addr.s_addr = htonl(INADDR_ANY);
        from.sin_port = htons(9001);                   // DiagManager.cpp:30; not fatal if somebody else holds it
        (void)::bind(fd, reinterpret_cast<sockaddr *>(&from), sizeof from);
        return true;
    }
    // demodulator.cpp:161-163: (float *)symbols, min(symbols, 1024) FLOATS
    void add(const float *iq, size_t nsym)
    {
        size_t nf = nsym < 1024 ? nsym : 1024;
        if (q.size() + nf > CAP) return;
        q.insert(q.end(), iq, iq + nf);
    }
    void pump()
    {
        const auto now = std::chrono::steady_clock::now();
        if (q.size() < 1024 || now - last < std::chrono::milliseconds(10)) return;
        char data[1024];
        for (int i = 0; i < 1024; ++i) {
            float v = q.front() * 128.f;
            q.pop_front();
            v = v > 127 ? 127 : v;
            v = v < -128 ? -128 : v;
            data[i] = static_cast<char>(v);
        }
        (void)::sendto(fd, data, sizeof data, 0, reinterpret_cast<const sockaddr *>(&to), sizeof to);
        last = now;
    }
};

}  // namespace

// --gpus N: one process drives N GPUs (ncclCommInitAll).  A block of the capture is cut in N time slices, one per
// GPU; the ranks exchange the halo and the boundary symbols over RCCL (xrit_group_process_slice_device) and the
// symbols are sent on in rank order.  Every block starts from cold chains (a rank's previous slice ended somewhere
// else in the stream), so blocks should be long: the halo is ~1 M input samples per rank at LRIT, decimation 5.
static int run_multi_gpu(const Options &o, int type, size_t bytes_per_sample, const xrit_demod_config &cfg0)
{
    const int W = o.gpus;
    if (xrit_device_count() < W) { std::fprintf(stderr, "--gpus %d: only %d HIP devices\n", W, xrit_device_count()); return 1; }
    std::vector<int> devs((size_t)W);
    for (int i = 0; i < W; ++i) devs[(size_t)i] = i;
    std::vector<xrit_group *> grp((size_t)W, nullptr);
    if (xrit_group_create_all(&cfg0, devs.data(), W, grp.data()) != XRIT_OK) {
        std::fprintf(stderr, "xritdemod_amd: %s\n", xrit_last_error());
        return 1;
    }
    FILE *in = std::fopen(o.input.c_str(), "rb");
    if (!in) { std::perror("input"); return 1; }
    Sink sink;
    sink.sndbuf = o.sndbuf;
    if (!sink.open(o.sink, o.connect_tries)) { std::fclose(in); return 1; }
    const size_t halo = xrit_group_halo_samples(grp[0]);
    const size_t per = o.block / (size_t)W / cfg0.decimation * cfg0.decimation;
    if (per < halo + 1024) { std::fprintf(stderr, "--block %zu is too short for %d slices with a halo of %zu samples\n", o.block, W, halo); return 2; }
    std::vector<unsigned char> raw(per * (size_t)W * bytes_per_sample);
    const size_t cap = per + 1024;
    std::vector<std::vector<float>> soft((size_t)W, std::vector<float>(cap));
    std::vector<size_t> count((size_t)W, 0);
    std::vector<uint64_t> offset((size_t)W, 0);
    std::vector<int> rcs((size_t)W, 0);
    std::vector<int8_t> q(cap);
    size_t total_in = 0, total_sym = 0;
    int exit_code = 0;
    const auto t_start = std::chrono::steady_clock::now();
    for (;;) {
        const size_t n = std::fread(raw.data(), bytes_per_sample, per * (size_t)W, in);
        if (n < per * (size_t)W) { std::fprintf(stderr, n ? "EOF (last %zu samples do not fill the slices: dropped)\n" : "EOF\n", n); break; }
        std::vector<std::thread> th;
        for (int r = 0; r < W; ++r)
            th.emplace_back([&, r] {
                // the slice goes to the rank's GPU through the chain's host entry point's twin: device buffers are the
                // group's business, so use the simplest route -- a device copy owned by this thread
                rcs[(size_t)r] = xrit_group_process_slice_host(grp[(size_t)r], raw.data() + (size_t)r * per * bytes_per_sample, per,
                                                               type, soft[(size_t)r].data(), cap, &count[(size_t)r],
                                                               &offset[(size_t)r], nullptr);
            });
        for (auto &t : th) t.join();
        for (int r = 0; r < W; ++r)
            if (rcs[(size_t)r] != XRIT_OK) { std::fprintf(stderr, "rank %d: %s\n", r, xrit_last_error()); exit_code = 1; }
        if (exit_code) break;
        for (int r = 0; r < W && !exit_code; ++r) {
            // SymbolManager.cpp:43-46 on the host: the ranks' pieces are already in stream order (offset[r] ascending)
            for (size_t i = 0; i < count[(size_t)r]; ++i) {
                float f = soft[(size_t)r][i] * 127;
                f = f > 127 ? 127 : f;
                f = f < -128 ? -128 : f;
                q[i] = (int8_t)(char)f;
            }
            if (!sink.send_all(q.data(), count[(size_t)r])) exit_code = 1;
            total_sym += count[(size_t)r];
        }
        total_in += n;
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    if (o.stats)
        std::fprintf(stderr, "samples in %zu, symbols out %zu, %.3f s (%.2f Msamples/s incl. file and PCIe), %d GPUs, halo %zu samples per boundary\n",
                     total_in, total_sym, secs, secs > 0 ? total_in / secs * 1e-6 : 0.0, W, halo);
    sink.close_all();
    std::fclose(in);
    for (auto g : grp) xrit_group_destroy(g);
    return exit_code;
}

int main(int argc, char **argv)
{
    Options o;
    if (!parse(argc, argv, o)) { usage(); return 2; }
    int type;
    size_t bytes_per_sample;
    if (o.format == "cf32") { type = XRIT_SAMPLE_FLOATIQ; bytes_per_sample = 8; }
    else if (o.format == "s16") { type = XRIT_SAMPLE_S16IQ; bytes_per_sample = 4; }
    else if (o.format == "s8") { type = XRIT_SAMPLE_S8IQ; bytes_per_sample = 2; }
    else if (o.format == "u8") { type = XRIT_SAMPLE_U8IQ; bytes_per_sample = 2; }      // raw rtl_sdr capture
    else { usage(); return 2; }

    xrit_demod_config cfg;
    if (o.mode == "hrit") xrit_demod_config_hrit(&cfg, (float)o.sample_rate, o.decimation);
    else xrit_demod_config_lrit(&cfg, (float)o.sample_rate, o.decimation);
    int rc = XRIT_OK;
    cfg.device = o.device;
    cfg.front_exact = o.front_exact;
    if (o.gpus > 1) return run_multi_gpu(o, type, bytes_per_sample, cfg);
    xrit_demod *chain = nullptr;
    if (xrit_demod_create(&cfg, &chain) != XRIT_OK) {
        // e.g. no HIP device: there is no CPU path
        std::fprintf(stderr, "xritdemod_amd: %s\n", xrit_last_error());
        return 1;
    }

    FILE *in = std::fopen(o.input.c_str(), "rb");
    if (!in) { std::perror("input"); xrit_demod_destroy(chain); return 1; }
    Sink sink;
    SymbolQueue queue;
    queue.cap = o.queue_symbols;
    sink.sndbuf = o.sndbuf;
    std::atomic<bool> dsp_done{false};
    std::thread sender;
    if (o.drop) {
        // the reference's symbolThread / process() loop: connect attempts happen HERE, next to the DSP, not before it
        if (o.sink.rfind("tcp://", 0) != 0) { std::fprintf(stderr, "--drop needs a tcp:// sink\n"); std::fclose(in); xrit_demod_destroy(chain); return 2; }
        const std::string hp = o.sink.substr(6);
        const size_t c = hp.rfind(':');
        if (c == std::string::npos) { std::fprintf(stderr, "sink: tcp://HOST:PORT expected\n"); std::fclose(in); xrit_demod_destroy(chain); return 2; }
        sink.kind = Sink::TCP;
        sink.host = hp.substr(0, c);
        sink.port = std::atoi(hp.c_str() + c + 1);
        sender = std::thread([&] {
            std::vector<int8_t> buf(16384);
            bool connected = false;
            int failed = 0;
            for (;;) {
                if (!connected) {
                    std::fprintf(stderr, "Trying to connect to decoder at %s:%d\n", sink.host.c_str(), sink.port);
                    connected = sink.connect_once();
                    if (!connected) {
                        queue.clear_disconnected();          // SymbolManager.cpp:78-83
                        if (dsp_done.load() && ++failed >= 2) return;
                        std::this_thread::sleep_for(std::chrono::seconds(1));
                        continue;
                    }
                }
                const size_t n = queue.take(buf.data(), buf.size());
                if (n == 0) {
                    if (dsp_done.load()) return;
                    std::this_thread::sleep_for(std::chrono::microseconds(200));
                    continue;
                }
                size_t off = 0;
                while (off < n) {
                    const ssize_t w = ::send(sink.fd, buf.data() + off, n - off, MSG_NOSIGNAL);
                    if (w <= 0) {
                        std::fprintf(stderr, "Disconnected from decoder.\n");
                        ::close(sink.fd);
                        sink.fd = -1;
                        connected = false;
                        break;                                 // the rest of this piece is lost, like the reference's
                    }
                    off += (size_t)w;
                }
                sink.sent += off;
            }
        });
    } else if (!sink.open(o.sink, o.connect_tries)) { std::fclose(in); xrit_demod_destroy(chain); return 1; }

    Diag diag;
    std::vector<float> diag_iq;
    if (!o.diag.empty()) {
        if (!diag.open(o.diag) || xrit_demod_keep_stages(chain, 2) != XRIT_OK) {
            std::fprintf(stderr, "diag: cannot set up %s\n", o.diag.c_str());
            sink.close_all(); std::fclose(in); xrit_demod_destroy(chain);
            return 1;
        }
        diag_iq.resize(2 * 1024);
    }
    const auto t_start = std::chrono::steady_clock::now();
    auto t_release = t_start;
    const size_t max_chunk = o.fifo ? FIFO_COMPLEX : o.block;
    std::vector<unsigned char> raw(max_chunk * bytes_per_sample);
    const size_t cap = max_chunk + 64;
    // --fifo: samplesFifo (demodulator.cpp:38) between the source's callbacks and the DSP loop
    size_t fifo_fill = 0, fifo_overflowed = 0, fifo_calls = 0, fifo_min = ~(size_t)0, fifo_max = 0;
    bool fifo_eof = false;
    std::vector<unsigned char> blockbuf(o.fifo ? o.fifo_block * bytes_per_sample : 0);
    auto fifo_chunk = [&]() -> size_t {
        // the source runs until the DSP loop looks again AND finds at least 64 Ki floats (:108-116)
        for (;;) {
            for (int b = 0; b < o.fifo_lag && !fifo_eof; ++b) {
                const size_t got = std::fread(blockbuf.data(), bytes_per_sample, o.fifo_block, in);
                if (got == 0) { fifo_eof = true; break; }
                if (o.paced) {        // CFileFrontend.cpp:36-47: one callback per block period
                    std::this_thread::sleep_until(t_release);
                    t_release += std::chrono::duration_cast<std::chrono::steady_clock::duration>(
                        std::chrono::duration<double>((double)got / o.sample_rate));
                }
                size_t take = got;
                if (fifo_fill + take > FIFO_COMPLEX) {
                    if (!fifo_overflowed) std::fprintf(stderr, "Input Samples Fifo is overflowing!\n");
                    fifo_overflowed += fifo_fill + take - FIFO_COMPLEX;
                    take = FIFO_COMPLEX - fifo_fill;
                }
                std::memcpy(raw.data() + fifo_fill * bytes_per_sample, blockbuf.data(), take * bytes_per_sample);
                fifo_fill += take;
            }
            if (fifo_fill >= FIFO_MIN_COMPLEX || fifo_eof) break;
        }
        // (at the end of the file the reference's loop stops with what is left below the threshold still in the FIFO)
        if (fifo_fill < FIFO_MIN_COMPLEX) return 0;
        const size_t n = fifo_fill;
        fifo_fill = 0;
        ++fifo_calls;
        fifo_min = n < fifo_min ? n : fifo_min;
        fifo_max = n > fifo_max ? n : fifo_max;
        return n;
    };
    std::vector<float> soft(cap);
    std::vector<int8_t> q(cap);
    size_t total_in = 0, total_sym = 0;
    int exit_code = 0;
    for (;;) {
        size_t n = o.fifo ? fifo_chunk() : std::fread(raw.data(), bytes_per_sample, o.block, in);
        if (n == 0) { std::fprintf(stderr, "EOF\n"); break; }
        if (o.paced && !o.fifo) {
            // CFileFrontend.cpp:36-47: one block per block period
            std::this_thread::sleep_until(t_release);
            t_release += std::chrono::duration_cast<std::chrono::steady_clock::duration>(
                std::chrono::duration<double>((double)n / o.sample_rate));
        }
        size_t nsym = 0;
        rc = xrit_demod_process(chain, raw.data(), n, type, soft.data(), cap, &nsym);
        if (rc != XRIT_OK) { std::fprintf(stderr, "process: %s\n", xrit_last_error()); exit_code = 1; break; }
        if (o.drop) {
            queue.add(soft.data(), nsym);         // SymbolManager::add: never waits, may drop
        } else {
            rc = xrit_quantize_i8(chain, soft.data(), q.data(), nsym);
            if (rc != XRIT_OK) { std::fprintf(stderr, "quantize: %s\n", xrit_last_error()); exit_code = 1; break; }
            if (!sink.send_all(q.data(), nsym)) { exit_code = 1; break; }
        }
        if (diag.fd >= 0 && nsym > 0) {
            // the first symbols of the call, complex (stage 4); only what the tap can take is copied back
            std::vector<float> all;
            size_t have = 0;
            if (xrit_demod_read_stage(chain, 4, nullptr, 0, &have) == XRIT_OK && have > 0) {
                all.resize(2 * have);
                if (xrit_demod_read_stage(chain, 4, all.data(), have, &have) == XRIT_OK) diag.add(all.data(), nsym);
            }
            diag.pump();
        }
        total_in += n;
        total_sym += nsym;
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    const size_t fill_at_eof = o.drop ? queue.size() : 0;
    if (o.drop) {
        dsp_done.store(true);
        sender.join();
    }
    if (o.stats) {
        xrit_demod_stats st;
        if (xrit_demod_get_stats(chain, &st) == XRIT_OK)
            std::fprintf(stderr, "samples in %zu, symbols out %zu, %.3f s (%.2f Msamples/s incl. file and PCIe)\n",
                         total_in, total_sym, secs, secs > 0 ? total_in / secs * 1e-6 : 0.0);
        if (o.fifo)
            std::fprintf(stderr, "fifo: %zu chunks of %zu .. %zu samples (source blocks of %zu, %d per look), %zu samples lost to overflow\n",
                         fifo_calls, fifo_calls ? fifo_min : 0, fifo_max, o.fifo_block, o.fifo_lag, fifo_overflowed);
        if (o.drop)
            std::fprintf(stderr, "symbol queue: capacity %zu, peak %zu, demodulatorFifoUsage %u %% at end of input (peak %u %%), "
                                 "sent %zu, dropped while full %zu, dropped while disconnected %zu\n",
                         queue.cap, queue.peak, SymbolQueue::usage(fill_at_eof, queue.cap),
                         SymbolQueue::usage(queue.peak, queue.cap), sink.sent, queue.dropped_full,
                         queue.dropped_disconnected);
    }
    sink.close_all();
    std::fclose(in);
    xrit_demod_destroy(chain);
    return exit_code;
}
