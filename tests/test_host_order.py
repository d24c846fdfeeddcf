"""agh_order_matches (agrep_amd/csrc/agh_order.h, host-only): the file order of a segment's match list from the
record numbers, with the start offsets as the referee -- compiled with g++ and checked against a plain sort."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <stdio.h>
#include <stdlib.h>
#include "agh_order.h"
static uint64_t rng = 88172645463325252ull;
static uint64_t next() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; }
int main()
{
    for (int trial = 0; trial < 200; ++trial) {
        const size_t n = trial < 3 ? (size_t)trial : (size_t)(next() % 5000) + 1;
        // records in file order: strictly increasing starts, increasing numbers; then shuffled
        std::vector<uint64_t> st(n);
        std::vector<uint32_t> rec(n);
        uint64_t pos = next() % 1000;
        uint32_t r = (uint32_t)(next() % 100);
        for (size_t i = 0; i < n; ++i) {
            st[i] = pos; rec[i] = r;
            pos += 1 + next() % 300; r += 1 + (uint32_t)(next() % 5);
        }
        if (trial % 7 == 3 && n > 4) r = 0xfffffff0u, rec[n - 1] = 0xffffffffu;      // the top of the range
        for (size_t i = n; i > 1; --i) {
            const size_t j = next() % i;
            std::swap(st[i - 1], st[j]); std::swap(rec[i - 1], rec[j]);
        }
        if (trial % 11 == 5 && n > 10)          // numbers that contradict the offsets: the offsets decide
            for (size_t i = 0; i < n; i += 3) rec[i] = (uint32_t)next();
        std::vector<uint32_t> order;
        agh_order_matches(rec.data(), st.data(), n, order);
        if (order.size() != n) { printf("size\n"); return 1; }
        std::vector<char> seen(n, 0);
        for (size_t i = 0; i < n; ++i) {
            if (order[i] >= n || seen[order[i]]) { printf("not a permutation (trial %d)\n", trial); return 1; }
            seen[order[i]] = 1;
            if (i && st[order[i]] < st[order[i - 1]]) { printf("not in file order (trial %d)\n", trial); return 1; }
        }
    }
    printf("ok\n");
    return 0;
}
'''


def test_match_list_comes_out_in_file_order(tmp_path):
    src = tmp_path / "order_test.cpp"
    src.write_text(SRC)
    exe = tmp_path / "order_test"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "agrep_amd", "csrc"), str(src), "-o", str(exe)])
    out = subprocess.run([str(exe)], stdout=subprocess.PIPE, timeout=60)
    assert out.returncode == 0 and out.stdout.strip() == b"ok", out.stdout
