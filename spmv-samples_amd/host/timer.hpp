// timer.hpp — host timer with the reference's four-call contract
// (reference include/timer.hpp:8-66): Timer::total_start/stop bracket the whole
// SpMV<>() call (spmv.h:38-40), Timer::kernel_start/stop bracket the device work
// inside a kind (cusp_warp_reduce.cuh:130-132), and the harness reads
// total_cost()/kernel_cost() after every call (main.cu:108-109).  Costs are whole
// microseconds, as in the reference (timer.hpp:10).  Single host thread, like the
// reference's singleton (timer.hpp:12-15).
#pragma once

#include <chrono>
#include <cstdint>

class Timer {
public:
    using clock = std::chrono::steady_clock;

    static void total_start() { slot().t[0] = clock::now(); }
    static void total_stop() { slot().t[1] = clock::now(); }
    static void kernel_start() { slot().k[0] = clock::now(); }
    static void kernel_stop() { slot().k[1] = clock::now(); }

    static int64_t total_cost() { return us(slot().t[0], slot().t[1]); }
    static int64_t kernel_cost() { return us(slot().k[0], slot().k[1]); }

    Timer(const Timer&) = delete;
    Timer& operator=(const Timer&) = delete;

private:
    Timer() = default;
    struct Slot {
        clock::time_point t[2]{}, k[2]{};
    };
    static Slot& slot() {
        static Slot s;
        return s;
    }
    static int64_t us(clock::time_point a, clock::time_point b) {
        return std::chrono::duration_cast<std::chrono::microseconds>(b - a).count();
    }
};
