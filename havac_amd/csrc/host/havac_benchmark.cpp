// havac_benchmark.cpp -- counterpart of the reference's benchmark/benchmark.cpp:25-81: wall-clock
// timing of the four stages of a run through the public `Havac` API.
//   usage: havac_benchmark <fasta> <hmm> [p-value] [--sync-start] [--repeat N [--depth D]]
//          havac_benchmark --raw <packed sequence> <int8 model> --repeat N [--depth D] [--no-readback]
// (the reference's first argument, the xclbin path, has no meaning here and is not taken)
// The device layer starts on a helper thread while the files are read (Havac::DeferredStart): "build" is then the
// constructor's return and the HIP start-up shows inside "load", where it overlaps the parsing; --sync-start uses the
// reference-shaped constructor, which returns when the device is ready.
//
// --repeat N (round 5): after the first run, the same run N more times with up to D runs in flight (--depth, default 2;
// Havac::setPipelineDepth) -- runHardwareClientAsync, and getHitsFromFinishedRun for the oldest run once D are open: the loop a
// caller makes who searches one database with one model file after the other -- and the rate over those N runs in GCUPS.  No
// Python in the process: this is what the drop-in C++ API reaches.  --raw takes the device layer's own inputs instead of files
// -- a 2-bit packed sequence (host/sequence/SequencePreprocessor.cpp:46-70; whole 12288-column segments) and an int8 [row][A,C,G,T]
// model, as bench.py's synthetic workloads are (tools/dump_workload.py writes them) -- and drives the C ABI directly
// (include/havac_dev.h: the entry points the reference's HavacHwClient maps onto), every run's list fetched in full
// (--no-readback: counted only, the records stay in HBM -- what bench.py's timed region does).
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/havac_dev.h"
#include "Havac.hpp"

using clock_type = std::chrono::high_resolution_clock;
static double us(clock_type::time_point a, clock_type::time_point b) {
    return (double)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count();
}

static std::vector<char> read_file(const std::string &path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) { std::cerr << "error: cannot read " << path << std::endl; std::exit(2); }
    std::vector<char> data((size_t)f.tellg());
    f.seekg(0);
    f.read(data.data(), (std::streamsize)data.size());
    return data;
}

// the C ABI directly: packed sequence + int8 model in, `repeat` runs with `depth` in flight
static int raw_runs(const std::string &seqPath, const std::string &modelPath, int repeat, int depth, bool readback) {
    const std::vector<char> seq = read_file(seqPath), model = read_file(modelPath);
    havac_dev *dev = nullptr;
    auto check = [&](int rc, const char *what) {
        if (rc < 0) { std::cerr << "error: " << what << ": " << (dev ? havac_dev_last_error(dev) : "no device") << std::endl; std::exit(1); }
    };
    check(havac_dev_create(0, &dev), "havac_dev_create");
    const uint64_t cells = (uint64_t)seq.size() * 4 * (model.size() / 4);
    check(havac_dev_set_hit_capacity(dev, std::max<uint64_t>(1u << 20, (uint64_t)((double)cells * 4e-5))), "havac_dev_set_hit_capacity");
    check(havac_dev_set_pipeline_depth(dev, (uint32_t)depth), "havac_dev_set_pipeline_depth");
    check(havac_dev_write_sequence(dev, reinterpret_cast<const uint8_t *>(seq.data()), seq.size()), "havac_dev_write_sequence");
    check(havac_dev_write_phmm(dev, reinterpret_cast<const int8_t *>(model.data()), model.size()), "havac_dev_write_phmm");
    std::vector<uint64_t> hits;
    uint64_t first_count = 0, first_sum = 0;
    bool same = true;
    auto fetch = [&]() {        // the oldest open run: wait, count, list, close
        check(havac_dev_wait(dev, 0), "havac_dev_wait");
        uint64_t n = 0;
        check(havac_dev_num_hits64(dev, &n), "havac_dev_num_hits64");
        hits.resize(readback ? n : 0);
        if (n && readback) check(havac_dev_read_hits64(dev, hits.data(), n), "havac_dev_read_hits64");
        uint64_t sum = 1;
        for (uint64_t v : hits) sum += v;
        if (first_count == 0 && first_sum == 0) { first_count = n; first_sum = sum; }
        else if (n != first_count || sum != first_sum) same = false;
        check(havac_dev_retire(dev), "havac_dev_retire");
    };
    for (int i = 0; i < 10 + depth; i++) {                     // the clocks come up, every slot has its buffers
        check(havac_dev_run_async(dev), "havac_dev_run_async");
        if ((int)havac_dev_open_runs(dev) == depth) fetch();
    }
    while (havac_dev_open_runs(dev)) fetch();
    const auto t0 = clock_type::now();
    for (int i = 0; i < repeat; i++) {
        check(havac_dev_run_async(dev), "havac_dev_run_async");
        if ((int)havac_dev_open_runs(dev) == depth) fetch();
    }
    while (havac_dev_open_runs(dev)) fetch();
    const auto t1 = clock_type::now();
    const double ms = us(t0, t1) / 1e3 / repeat;
    std::cout << "raw inputs: " << seq.size() * 4 << " columns x " << model.size() / 4 << " rows, " << first_count << " hits per run, the same list every run: "
              << (same ? "yes" : "NO") << std::endl;
    std::cout << repeat << " runs, " << depth << " in flight, " << (readback ? "every list read back" : "lists left on the device (as bench.py's timed region)") << ": " << ms << " ms per run = " << (double)cells / ms / 1e6 << " GCUPS" << std::endl;
    havac_dev_destroy(dev);
    return same ? 0 : 1;
}

int main(int argc, char **argv) {
    std::srand((unsigned)std::time(nullptr));   // as the reference does (benchmark/benchmark.cpp:27)
    bool syncStart = false, raw = false, readback = true;
    int repeat = 0, depth = 2;
    std::vector<std::string> args;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--sync-start") syncStart = true;
        else if (a == "--raw") raw = true;
        else if (a == "--no-readback") readback = false;
        else if (a == "--repeat" && i + 1 < argc) repeat = std::atoi(argv[++i]);
        else if (a == "--depth" && i + 1 < argc) depth = std::atoi(argv[++i]);
        else args.push_back(a);
    }
    if (args.size() < 2 || depth < 1 || depth > 4 || repeat < 0 || (raw && repeat == 0)) {
        std::cout << "error: program requires the fasta file src and the hmm src  (or: --raw <packed sequence> <int8 model> --repeat N [--depth D])" << std::endl;
        return 2;
    }
    if (raw) return raw_runs(args[0], args[1], repeat, depth, readback);
    const float pValue = args.size() > 2 ? std::strtof(args[2].c_str(), nullptr) : 0.02f;
    auto t0 = clock_type::now();
    auto havac = syncStart ? std::make_shared<Havac>(0, pValue) : std::make_shared<Havac>(Havac::DeferredStart{}, 0, pValue);
    auto t1 = clock_type::now();
    havac->loadPhmm(args[1]);
    havac->loadSequence(args[0]);
    auto t2 = clock_type::now();
    havac->runHardwareClient();
    auto t3 = clock_type::now();
    vector<HavacHit> hits = havac->getHitsFromFinishedRun();
    auto t4 = clock_type::now();
    float kernelMs = 0, totalMs = 0;
    havac->lastRunMilliseconds(&kernelMs, &totalMs);
    std::cout << "hw generated " << hits.size() << " verified hits." << std::endl;
    std::cout << "timing information:" << std::endl;
    std::cout << "havac build time " << us(t0, t1) << " microseconds (" << us(t0, t1) / 1e6 << " seconds)." << std::endl;
    std::cout << "havac load time " << us(t1, t2) << " microseconds (" << us(t1, t2) / 1e6 << " seconds)." << std::endl;
    std::cout << "havac run time " << us(t2, t3) << " microseconds (" << us(t2, t3) / 1e6 << " seconds)." << std::endl;
    std::cout << "havac verify time " << us(t3, t4) << " microseconds (" << us(t3, t4) / 1e6 << " seconds)." << std::endl;
    std::cout << "total time taken " << us(t0, t4) << " microseconds (" << us(t0, t4) / 1e6 << " seconds)." << std::endl;
    std::cout << "device time: ssv kernel " << kernelMs << " ms, enqueue to ordered hits " << totalMs << " ms" << std::endl;
    if (repeat > 0) {
        // the same run again and again, `depth` in flight: every run's hits fetched and resolved, as a caller's loop would
        havac->setPipelineDepth((uint32_t)depth);
        size_t open = 0, resolved = 0;
        bool same = true;
        auto fetch = [&]() { vector<HavacHit> h = havac->getHitsFromFinishedRun(); resolved += h.size(); same = same && h.size() == hits.size(); open--; };
        for (int i = 0; i < 4 + depth; i++) { havac->runHardwareClientAsync(); if (++open == (size_t)depth) fetch(); }
        while (open) fetch();
        resolved = 0;
        const auto r0 = clock_type::now();
        for (int i = 0; i < repeat; i++) { havac->runHardwareClientAsync(); if (++open == (size_t)depth) fetch(); }
        while (open) fetch();
        const auto r1 = clock_type::now();
        std::cout << repeat << " more runs, " << depth << " in flight, hits fetched and resolved every run (" << resolved / (size_t)repeat << " each, the same count every run: "
                  << (same ? "yes" : "NO") << "): " << us(r0, r1) / 1e3 / repeat << " ms per run" << std::endl;
        if (!same) return 1;
    }
    return 0;
}
