// agh_order.h -- host-side ordering of a segment's match list (no HIP dependency: tests/test_host_order.py
// compiles it with g++).  The numbered pipeline appends matches in the order its waves finish; output() wants
// file order (agrep.c:3805-3956).  Records are numbered in file order, so the record number IS the sort key:
// packed with the list index into one 64-bit word, a plain std::sort of words replaces the indirect
// comparison sort on start offsets (5 ms for the 104 197 records of BASELINE config 2 -- most of what
// agh_scan_device_emit spent outside the scan).  Should the numbers ever not be monotone in the offsets
// (they are by construction), the offsets decide.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

inline void agh_order_matches(const uint32_t *rec, const uint64_t *start, size_t n, std::vector<uint32_t> &order)
{
    order.resize(n);
    if (n >= ((size_t)1 << 32)) {               // (the index does not fit the low half: the plain way)
        std::vector<size_t> o(n);
        for (size_t i = 0; i < n; ++i) o[i] = i;
        std::sort(o.begin(), o.end(), [&](size_t a, size_t b) { return start[a] < start[b]; });
        for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)o[i];
        return;
    }
    std::vector<uint64_t> key(n);
    for (size_t i = 0; i < n; ++i) key[i] = (uint64_t)rec[i] << 32 | (uint64_t)i;
    std::sort(key.begin(), key.end());
    bool monotone = true;
    for (size_t i = 0; i < n; ++i) {
        order[i] = (uint32_t)key[i];
        if (i && start[order[i]] < start[order[i - 1]]) monotone = false;
    }
    if (!monotone)
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return start[a] < start[b]; });
}
