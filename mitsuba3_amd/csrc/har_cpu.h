/* Host cores this process may really use: the scheduler affinity mask (what util::core_count() of the reference counts, src/core/util.cpp) capped by the
 * container's CPU bandwidth quota (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us / cpu.cfs_period_us`).  A GPU box of this pool shows 256 logical CPUs with a
 * quota of 16: 256 worker threads there run at an eighth of the rate of 16 (measured, tools/cpu_scaling.py). */
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <sched.h>

inline unsigned har_usable_cores() {
    static const unsigned cached = [] {
        unsigned n = std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n, (unsigned) std::max(1, CPU_COUNT(&set)));
        double quota = -1.0, period = -1.0;
        if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {                              /* "max 100000" or "<quota> <period>" */
            char q[32] = { 0 };
            if (std::fscanf(f, "%31s %lf", q, &period) == 2 && q[0] != 'm') quota = std::atof(q);
            std::fclose(f);
        } else {
            long long qv = -1, pv = -1;
            if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(g, "%lld", &qv) != 1) qv = -1; std::fclose(g); }
            if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(g, "%lld", &pv) != 1) pv = -1; std::fclose(g); }
            quota = (double) qv; period = (double) pv;
        }
        if (quota > 0.0 && period > 0.0) n = std::min(n, (unsigned) std::max(1.0, std::ceil(quota / period)));
        return n;
    }();
    return cached;
}
