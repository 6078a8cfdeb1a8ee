// havac_benchmark.cpp -- counterpart of the reference's benchmark/benchmark.cpp:25-81: wall-clock
// timing of the four stages of a run through the public `Havac` API.
//   usage: havac_benchmark <fasta> <hmm> [p-value] [--sync-start]
// (the reference's first argument, the xclbin path, has no meaning here and is not taken)
// The device layer starts on a helper thread while the files are read (Havac::DeferredStart): "build" is then the
// constructor's return and the HIP start-up shows inside "load", where it overlaps the parsing; --sync-start uses the
// reference-shaped constructor, which returns when the device is ready.
#include <chrono>
#include <cstdlib>
#include <ctime>
#include <iostream>
#include <memory>
#include <string>

#include "Havac.hpp"

int main(int argc, char **argv) {
    std::srand((unsigned)std::time(nullptr));   // as the reference does (benchmark/benchmark.cpp:27)
    if (argc < 3) {
        std::cout << "error: program requires the fasta file src and the hmm src" << std::endl;
        return 2;
    }
    bool syncStart = false;
    for (int i = 3; i < argc; i++)
        if (std::string(argv[i]) == "--sync-start") { syncStart = true; for (int j = i; j + 1 < argc; j++) argv[j] = argv[j + 1]; argc--; i--; }
    const float pValue = argc > 3 ? std::strtof(argv[3], nullptr) : 0.02f;
    using clock = std::chrono::high_resolution_clock;
    auto us = [](clock::time_point a, clock::time_point b) {
        return (double)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count();
    };
    auto t0 = clock::now();
    auto havac = syncStart ? std::make_shared<Havac>(0, pValue) : std::make_shared<Havac>(Havac::DeferredStart{}, 0, pValue);
    auto t1 = clock::now();
    havac->loadPhmm(argv[2]);
    havac->loadSequence(argv[1]);
    auto t2 = clock::now();
    havac->runHardwareClient();
    auto t3 = clock::now();
    vector<HavacHit> hits = havac->getHitsFromFinishedRun();
    auto t4 = clock::now();
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
    return 0;
}
