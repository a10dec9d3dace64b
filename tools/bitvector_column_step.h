// bitvector_column_step.h -- kept for the lanes = reads kernel of DESIGN.md section 9 (not part of the product build).
//
// One column step in the reference's own bit-vector form (getNextSlice, GraphAligner.h:1349-1427, without the row
// confirmation that only cycles need) plus the vertical re-entry (calculateNode :1541-1546) without a general merge.
// It was written for an experiment that ran the column loop of part of the waves on the scalar unit (bit-exact in the
// host emulation and on the MI355X, but slower: the scalar unit retires one instruction per 4 cycles per SIMD); per LANE
// with 64-bit vector operations it is the inner loop measured in tools/ubench_lane_per_read.hip.
//
//   vp, vn   vertical +1 / -1 deltas of the column to the left (bit i = row j+i against row j+i-1); on return: of this column
//   before   scoreBeforeStart (row j-1) of the column to the left; on return: of this column
//   calc     scoreBeforeStart of this column before the re-entry test: min(left + 1, diagonal from the slice above)
//   flags    base (bits 0-1) | "no diagonal into row j" << 2 (the cell above the left column does not exist, :1358,1360)
//            | d << 3, d = calc - (score of the cell above this column) when the column is re-entered from above, else 0
//   e0..e3   match words of the slice's 64 read rows against A, C, G, T
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define GA_BV_FN __host__ __device__ static inline
#else
#define GA_BV_FN static inline
#endif

GA_BV_FN void bitvector_column_step(uint64_t& vp, uint64_t& vn, int& before, int calc, int flags, uint64_t e0, uint64_t e1, uint64_t e2, uint64_t e3)
{
	uint64_t eq = (flags & 2) ? ((flags & 1) ? e3 : e2) : ((flags & 1) ? e1 : e0);
	eq &= ~(uint64_t)((uint32_t)(flags >> 2) & 1u);
	const int hin = calc - before;                                              // -1, 0 or +1
	const uint64_t neg = (uint32_t)hin >> 31, pos = (uint32_t)(-hin) >> 31;     // sign tests as bits: no branches in the chain
	const uint64_t xv = eq | vn;
	eq |= neg;
	const uint64_t xh = (((eq & vp) + vp) ^ vp) | eq;
	uint64_t ph = vn | ~(xh | vp);
	uint64_t mh = vp & xh;
	ph = (ph << 1) | pos;
	mh = (mh << 1) | neg;
	vp = mh | ~(xv | ph);
	vn = ph & xv;
	before = calc;
	const int d = flags >> 3;
	if (d)
	{
		// cell-wise minimum with the vertical run coming down from the cell above, whose score is before - d
		// (mergeTwoSlices with a {VP = ~0, VN = 0} column, WordSlice.h:361-421).  With T_r = S_r - r the run is the
		// constant K = before - d + 1 and the column falls by one unit at every row without VP and by one more at
		// every row with VN: the run wins down to the row where d units have been lost.  Inside a node d <= 2
		// (adjacent cells of a row differ by at most one).
		int lost = 0, p = 64;
		bool exact = false;
		uint64_t m = ~vp;
		while (m)
		{
			const int i = __builtin_ctzll(m);
			m &= m - 1;
			lost += 1 + (int)((vn >> i) & 1);
			if (lost >= d) { p = i; exact = lost == d; break; }
		}
		const uint64_t below = p >= 64 ? ~0ull : ((1ull << p) - 1);
		const uint64_t keep = p >= 63 ? 0ull : (~0ull << (p + 1));
		vp = below | (vp & keep) | ((p < 64 && exact) ? (1ull << p) : 0ull);
		vn = vn & keep;
		before -= d;
	}
}

// Cell-wise minimum of two columns over rows j-1 .. j+63 (mergeTwoSlices, WordSlice.h:361-421), row by row: what a node
// start needs when the node has several in-neighbours in the band.  (The reference does this with byte-wise prefix sums in
// O(log w); at one merge per node a 64-step loop per lane is affordable, and it is the form that was checked.)
GA_BV_FN void bitvector_column_merge(uint64_t& vp, uint64_t& vn, int& before, uint64_t vp2, uint64_t vn2, int before2)
{
	int sa = before, sb = before2;
	int prev = sa < sb ? sa : sb;
	const int outBefore = prev;
	uint64_t ovp = 0, ovn = 0;
	for (int r = 0; r < 64; r++)
	{
		sa += (int)((vp >> r) & 1) - (int)((vn >> r) & 1);
		sb += (int)((vp2 >> r) & 1) - (int)((vn2 >> r) & 1);
		const int m = sa < sb ? sa : sb;
		ovp |= (uint64_t)(m == prev + 1) << r;
		ovn |= (uint64_t)(m == prev - 1) << r;
		prev = m;
	}
	vp = ovp; vn = ovn; before = outBefore;
}

// score of row j+row of a column (WordSlice::getValue, WordSlice.h:223-229)
GA_BV_FN int bitvector_column_value(uint64_t vp, uint64_t vn, int before, int row)
{
	const uint64_t mask = row < 63 ? ~(~0ull << (row + 1)) : ~0ull;
	return before + __builtin_popcountll(vp & mask) - __builtin_popcountll(vn & mask);
}
