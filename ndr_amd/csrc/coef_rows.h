// Coefficient rows of the per-class node rows (l1_merged_core.h) through the scalar cache, one row AHEAD of their use.
//
// A row is the nine signed entries of cK0[0] (level 1) or K0 (level 0) for one (mirror class, neighbour kind), 96-byte slots of
// l1m::build_table.  A wave walks its rows in a fixed order; with the request and the wait in one place (sload12) the ~200-cycle
// round trip of every row is exposed -- more than the ~45 multiply-adds between two rows -- so the request for row r + 1 is
// issued when row r is taken, into the other of two register sets.  A request in flight across compiler-generated code is only
// safe while the allocator neither spills nor copies its destination registers, and never across a branch: each role primes its
// pipeline inside its own branch, and tools/check_sload_pipeline.py scans the ISA of every build for instructions that touch the
// destination registers between a request and its wait (__graft_entry__.build, tests/test_abi_and_host.py).
#pragma once
#include "device_utils.h"
#include "l1_merged_core.h"

namespace vfem {

template <int OFF>
__device__ __forceinline__ void srow_issue(const double *p, d8_t &a, double &b) {
    asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx2 %1, %2, %4" : "=&s"(a), "=&s"(b) : "s"(p), "n"(OFF), "n"(OFF + 64));
}
__device__ __forceinline__ void srow_wait(d8_t &a, double &b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b)); }
// the same, tied to a value of the surrounding arithmetic: the compiler keeps asm statements in order among themselves but moves
// them freely past ordinary instructions -- where the operands of the arithmetic arrive late (LDS reads) it ran ALL requests and
// waits of a node up front and parked the rows in VGPR lanes (691 v_writelane).  `tie` is an accumulator the previous row's
// arithmetic wrote and the next row's reads: the wait can then stand only between the two.
__device__ __forceinline__ void srow_wait(d8_t &a, double &b, double &tie) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b), "+v"(tie)); }

// position of neighbour kind w in the row order of a part: side parts w = 7, 6, 5, 4; the middle part w = 3, 2, 0, 1
template <bool MID> constexpr int row_pos(int w) { return MID ? (w == 3 ? 0 : w == 2 ? 1 : w == 0 ? 2 : 3) : 7 - w; }
template <bool MID> constexpr int row_kind(int pos) { return MID ? (pos == 0 ? 3 : pos == 1 ? 2 : pos == 2 ? 0 : 1) : 7 - pos; }

// NROWS rows in the order SEQ::at(0), SEQ::at(1), ... (byte offsets into the table)
template <int NROWS, class SEQ>
struct RowPipe {
    const double *tab;
    d8_t a0, a1;
    double b0, b1;
    __device__ __forceinline__ void prime() { srow_issue<SEQ::at(0)>(tab, a0, b0); }
    template <int R>
    __device__ __forceinline__ void take(double c[9], double &tie) {
        static_assert(R >= 0 && R < NROWS, "row outside the sequence");
        if constexpr (R % 2 == 0) {
            srow_wait(a0, b0, tie);
            if constexpr (R + 1 < NROWS) srow_issue<SEQ::at(R + 1 < NROWS ? R + 1 : 0)>(tab, a1, b1);
            c[0] = a0[0]; c[1] = a0[1]; c[2] = a0[2]; c[3] = a0[3]; c[4] = a0[4]; c[5] = a0[5]; c[6] = a0[6]; c[7] = a0[7]; c[8] = b0;
        } else {
            srow_wait(a1, b1, tie);
            if constexpr (R + 1 < NROWS) srow_issue<SEQ::at(R + 1 < NROWS ? R + 1 : 0)>(tab, a0, b0);
            c[0] = a1[0]; c[1] = a1[1]; c[2] = a1[2]; c[3] = a1[3]; c[4] = a1[4]; c[5] = a1[5]; c[6] = a1[6]; c[7] = a1[7]; c[8] = b1;
        }
    }
    template <int R>
    __device__ __forceinline__ void take(double c[9]) {
        static_assert(R >= 0 && R < NROWS, "row outside the sequence");
        if constexpr (R % 2 == 0) {
            srow_wait(a0, b0);
            if constexpr (R + 1 < NROWS) srow_issue<SEQ::at(R + 1 < NROWS ? R + 1 : 0)>(tab, a1, b1);
            c[0] = a0[0]; c[1] = a0[1]; c[2] = a0[2]; c[3] = a0[3]; c[4] = a0[4]; c[5] = a0[5]; c[6] = a0[6]; c[7] = a0[7]; c[8] = b0;
        } else {
            srow_wait(a1, b1);
            if constexpr (R + 1 < NROWS) srow_issue<SEQ::at(R + 1 < NROWS ? R + 1 : 0)>(tab, a0, b0);
            c[0] = a1[0]; c[1] = a1[1]; c[2] = a1[2]; c[3] = a1[3]; c[4] = a1[4]; c[5] = a1[5]; c[6] = a1[6]; c[7] = a1[7]; c[8] = b1;
        }
    }
};

// level 1: the 32 rows of one part (all eight classes): side parts w = 7, 6, 5, 4 per class, the middle part w = 3, 2, 0, 1
template <bool MID>
struct L1PartRows { static constexpr int at(int r) { return ((r / 4) * 8 + row_kind<MID>(r % 4)) * l1m::TAB_ROW * 8; } };
template <bool MID>
struct DevCoef {                      // the `Coef` of l1m::side_class / mid_class
    RowPipe<32, L1PartRows<MID>> pipe;
    __device__ __forceinline__ explicit DevCoef(const double *tab) : pipe{tab} {}
    __device__ __forceinline__ void prime() { pipe.prime(); }
    template <int G, int W>
    __device__ __forceinline__ void get(double c[9]) { pipe.template take<G * 4 + row_pos<MID>(W)>(c); }
};

// level 0 (one class, g = 0): the twelve rows of a node in the order plane below (7, 6, 5, 4), own plane (3, 2, 0, 1), plane above
struct L0NodeRows {
    static constexpr int at(int r) { return (r < 4 ? 7 - r : r < 8 ? row_kind<true>(r - 4) : 7 - (r - 8)) * l1m::TAB_ROW * 8; }
};
// adapter of one part onto the node's pipeline; tie: the accumulator the rows' arithmetic chains through; after_first(position): called
// right behind the wait for each of the part's four rows -- the places to issue LDS reads: LDS and scalar loads share one counter, and a
// scalar load can only be awaited with lgkmcnt(0), so whatever LDS read is in flight at a row wait is waited for as well; a batch issued
// behind a row wait has that row's ~33 multiply-adds to land in
template <bool MID, int BASE, class Hook>
struct L0Coef {
    RowPipe<12, L0NodeRows> &pipe;
    double &tie;
    Hook after_first;
    template <int G, int W>
    __device__ __forceinline__ void get(double c[9]) {
        static_assert(G == 0, "level 0 has one class");
        pipe.template take<BASE + row_pos<MID>(W)>(c, tie);
        after_first(std::integral_constant<int, row_pos<MID>(W)>{});       // (called behind every row wait, with the row's position in the part)
    }
};
template <bool MID, int BASE, class Hook>
__device__ __forceinline__ L0Coef<MID, BASE, Hook> l0_coef(RowPipe<12, L0NodeRows> &pipe, double &tie, Hook hook) { return L0Coef<MID, BASE, Hook>{pipe, tie, hook}; }


}  // namespace vfem
