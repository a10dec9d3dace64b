// Self-check of tools/bitvector_column_step.h against the plain cell recurrence
//   S'(r) = min( S(r) + 1, S(r-1) + (match(r) ? 0 : 1), S'(r-1) + 1 ),   S'(-1) = scoreBeforeStart of the new column,
// on random valid columns, including the vertical re-entry (new scoreBeforeStart below what the row gives).
#include <cstdio>
#include <cstdlib>
#include <random>
#include "../../tools/bitvector_column_step.h"

int main()
{
	std::mt19937_64 rng(12345);
	long checked = 0, reentries = 0;
	for (int trial = 0; trial < 200000; trial++)
	{
		// a valid left column: scores with vertical deltas in {-1, 0, +1}
		int S[65];                                  // S[0] = row j-1 (before), S[r+1] = row j+r
		S[0] = 100 + (int)(rng() % 50);
		uint64_t vp = 0, vn = 0;
		for (int r = 0; r < 64; r++)
		{
			int d = (int)(rng() % 3) - 1;
			S[r + 1] = S[r] + d;
			if (d == 1) vp |= 1ull << r; else if (d == -1) vn |= 1ull << r;
		}
		const uint64_t eq = rng() & rng();          // sparse matches
		const bool noDiag = (rng() % 4) == 0;       // the cell above the left column does not exist: no diagonal into row j
		// scoreBeforeStart of the new column along the row: left + 1, or one less / equal through the diagonal of the slice above
		const int calc = S[0] + (int)(rng() % 3) - 1;   // hin in {-1, 0, +1}
		// cell above this column (previous slice's end score): sometimes lower than calc -> re-entry with d = 1 or 2
		int d = 0;
		if (rng() % 3 == 0) d = 1 + (int)(rng() % 2);
		const int beforeNew = calc - d;
		// expected column by the cell recurrence.  Row j's diagonal predecessor is S[0] (left column, row j-1); the horizontal
		// relation of row j-1 itself is what `calc` encodes, and a re-entered column starts d lower
		int E[65];
		E[0] = beforeNew;
		for (int r = 0; r < 64; r++)
		{
			bool match = (eq >> r) & 1;
			if (r == 0 && noDiag) match = false;
			int best = S[r + 1] + 1;
			int diag = S[r] + (match ? 0 : 1);
			// Myers' recurrence sees the row above through hin: for r == 0 the diagonal source is the left column's row j-1
			if (diag < best) best = diag;
			if (E[r] + 1 < best) best = E[r] + 1;
			E[r + 1] = best;
		}
		// the bit-vector step computes the column for scoreBeforeStart = calc and then takes the cell-wise minimum with the run from above;
		// the plain recurrence above with E[0] = beforeNew is that same minimum (min distributes over the recurrence)
		uint64_t a = vp, b = vn;
		int before = S[0];
		const int base = (int)(rng() % 4);
		uint64_t e[4] = {rng(), rng(), rng(), rng()};
		e[base] = eq;
		bitvector_column_step(a, b, before, calc, base | (noDiag ? 4 : 0) | (d << 3), e[0], e[1], e[2], e[3]);
		if (before != beforeNew) { printf("before mismatch trial %d\n", trial); return 1; }
		if (a & b) { printf("vp & vn overlap trial %d\n", trial); return 1; }
		int v = before;
		for (int r = 0; r < 64; r++)
		{
			v += (int)((a >> r) & 1) - (int)((b >> r) & 1);
			if (v != E[r + 1]) { printf("row %d mismatch trial %d: got %d want %d (d %d hin %d)\n", r, trial, v, E[r + 1], d, calc - S[0]); return 1; }
		}
		checked++;
		reentries += d > 0;
	}
	printf("bitvector_column_step: %ld random columns agree with the cell recurrence (%ld with re-entry)\n", checked, reentries);
	// ---- merge and value: two random valid columns, the merged words must encode the row-wise minimum ----
	long merged = 0;
	for (int trial = 0; trial < 100000; trial++)
	{
		int A[65], B[65];
		uint64_t avp = 0, avn = 0, bvp = 0, bvn = 0;
		A[0] = 50 + (int)(rng() % 20);
		B[0] = 50 + (int)(rng() % 20);
		for (int r = 0; r < 64; r++)
		{
			int d = (int)(rng() % 3) - 1, e = (int)(rng() % 3) - 1;
			A[r + 1] = A[r] + d; B[r + 1] = B[r] + e;
			if (d == 1) avp |= 1ull << r; else if (d == -1) avn |= 1ull << r;
			if (e == 1) bvp |= 1ull << r; else if (e == -1) bvn |= 1ull << r;
		}
		uint64_t vp = avp, vn = avn;
		int before = A[0];
		bitvector_column_merge(vp, vn, before, bvp, bvn, B[0]);
		if (before != (A[0] < B[0] ? A[0] : B[0]) || (vp & vn)) { printf("merge: before / overlap mismatch trial %d\n", trial); return 1; }
		for (int r = 0; r < 64; r++)
		{
			const int want = A[r + 1] < B[r + 1] ? A[r + 1] : B[r + 1];
			if (bitvector_column_value(vp, vn, before, r) != want) { printf("merge: row %d mismatch trial %d\n", r, trial); return 1; }
		}
		merged++;
	}
	printf("bitvector_column_merge / value: %ld random pairs agree with the row-wise minimum\n", merged);
	return 0;
}
