// ga_oracle.cpp -- see ga_oracle.hpp for scope, parity status and usage rules.
// TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED end-to-end (component pins: oracle/refparts.cpp).
#include "ga_oracle.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <queue>
#include <unordered_set>

namespace gao {

namespace {

constexpr int W = 64;                                  // WordSlice.h:13
constexpr size_t kCutoff = 200000;                     // GraphAlignerCommon.h:10,15
constexpr u64 kOnes = ~0ull;

[[noreturn]] void fail(Status s, const std::string& what) { throw Failure{s, what}; }
#define GAO_CHECK(cond) do { if (!(cond)) fail(ASSERTION, std::string(#cond) + " @" + std::to_string(__LINE__)); } while (0)

inline int pop(u64 x) { return __builtin_popcountll(x); }
inline int bitAt(u64 x, int i) { return (int)((x >> i) & 1); }

}  // namespace

// ===========================================================================================
// graph
// ===========================================================================================

Graph::Graph()
{
	// dummy start node occupies column 0 (AlignmentGraph.cpp:22-30)
	ids.push_back(0);
	start.push_back(0);
	in.emplace_back();
	out.emplace_back();
	rev.push_back(false);
	bases.push_back('-');
}

void Graph::addNode(int id, const std::string& seq, bool reverseNode)
{
	GAO_CHECK(!finalized);
	if (lookup.count(id) != 0) return;                 // duplicates are ignored (:51)
	lookup[id] = start.size();
	ids.push_back(id);
	start.push_back(bases.size());
	in.emplace_back();
	out.emplace_back();
	rev.push_back(reverseNode);
	for (char c : seq)
	{
		if (c != 'A' && c != 'C' && c != 'G' && c != 'T') fail(ASSERTION, "graph base outside ACGT");   // :83-85
		bases.push_back(c);
	}
}

void Graph::addEdge(int from, int to)
{
	GAO_CHECK(!finalized);
	GAO_CHECK(lookup.count(from) > 0);
	GAO_CHECK(lookup.count(to) > 0);
	size_t f = lookup[from], t = lookup[to];
	if (std::find(in[t].begin(), in[t].end(), f) == in[t].end()) in[t].push_back(f);        // :104
	if (std::find(out[f].begin(), out[f].end(), t) == out[f].end()) out[f].push_back(t);    // :105
}

void Graph::finalize()
{
	dummyLast = bases.size();
	ids.push_back(0);
	start.push_back(bases.size());
	rev.push_back(false);
	in.emplace_back();
	out.emplace_back();
	bases.push_back('-');
	dummyFirst = 0;
	finalized = true;
}

void Graph::addBigraphNode(int id, const std::string& seq)
{
	// BigraphToDigraph.cpp:27-30, 115-116 (vg) -- forward copy = 2*id, reverse complement = 2*id+1
	addNode(id * 2, seq, false);
	addNode(id * 2 + 1, reverseComplement(seq), true);
}

void Graph::addBigraphEdge(int from, bool fromStart, int to, bool toEnd)
{
	// BigraphToDigraph.cpp:32-56
	size_t fromLeft = fromStart ? from * 2 : from * 2 + 1;
	size_t fromRight = fromStart ? from * 2 + 1 : from * 2;
	size_t toLeft = toEnd ? to * 2 : to * 2 + 1;
	size_t toRight = toEnd ? to * 2 + 1 : to * 2;
	addEdge((int)fromRight, (int)toRight);
	addEdge((int)toLeft, (int)fromLeft);
}

size_t Graph::nodeOf(size_t column) const
{
	GAO_CHECK(column < bases.size());
	auto it = std::upper_bound(start.begin(), start.end(), column);
	return (size_t)(it - start.begin()) - 1;
}

char Graph::base(size_t column) const
{
	GAO_CHECK(column < bases.size());
	return bases[column];
}

size_t Graph::reverseNode(size_t n) const
{
	int big = ids[n] / 2;
	auto it = lookup.find(ids[n] % 2 == 1 ? big * 2 : big * 2 + 1);
	if (it == lookup.end()) fail(BAD_SEED, "reverse node missing");
	GAO_CHECK(it->second != n);
	GAO_CHECK(nodeLen(it->second) == nodeLen(n));
	return it->second;
}

size_t Graph::reverseColumn(size_t column) const
{
	GAO_CHECK(column < bases.size());
	GAO_CHECK(column > 0);
	size_t n = nodeOf(column);
	size_t o = reverseNode(n);
	return (nodeEnd(o) - 1) - (column - start[n]);
}

// ===========================================================================================
// characters
// ===========================================================================================

bool charMatch(char r, char gch)
{
	// GraphAligner.h:2039-2110.  The graph side must be ACGT (:2041).
	if (gch != 'A' && gch != 'C' && gch != 'G' && gch != 'T') fail(ASSERTION, "graph char not ACGT");
	const bool a = gch == 'A', c = gch == 'C', g = gch == 'G', t = gch == 'T';
	switch (r)
	{
		case 'A': case 'a': return a;
		case 'T': case 't': return t;
		case 'C': case 'c': return c;
		case 'G': case 'g': return g;
		case 'N': case 'n': return true;
		case 'R': case 'r': return a || g;
		case 'Y': case 'y': return c || t;
		case 'K': case 'k': return g || t;
		case 'M': case 'm': return c || a;
		case 'S': case 's': return c || g;
		case 'W': case 'w': return a || t;
		case 'B': case 'b': return c || g || t;
		case 'D': case 'd': return a || g || t;
		case 'H': case 'h': return a || c || t;
		case 'V': case 'v': return a || c || g;
		default: fail(ASSERTION, "read char not IUPAC");
	}
}

std::string reverseComplement(const std::string& s)
{
	// CommonUtils.cpp:60-136.  Note 'H'/'h' has no break in the reference (:128-132) and so
	// reaches the default branch's assert(false); 'U' maps to 'A'.
	std::string out;
	out.reserve(s.size());
	for (size_t k = s.size(); k-- > 0;)
	{
		char c;
		switch (s[k])
		{
			case 'A': case 'a': c = 'T'; break;
			case 'C': case 'c': c = 'G'; break;
			case 'T': case 't': c = 'A'; break;
			case 'G': case 'g': c = 'C'; break;
			case 'N': case 'n': c = 'N'; break;
			case 'U': case 'u': c = 'A'; break;
			case 'R': case 'r': c = 'Y'; break;
			case 'Y': case 'y': c = 'R'; break;
			case 'K': case 'k': c = 'M'; break;
			case 'M': case 'm': c = 'K'; break;
			case 'S': case 's': c = 'S'; break;
			case 'W': case 'w': c = 'W'; break;
			case 'B': case 'b': c = 'V'; break;
			case 'V': case 'v': c = 'B'; break;
			case 'D': case 'd': c = 'H'; break;
			default: fail(ASSERTION, "reverse complement of unsupported char");   // includes 'H'/'h'
		}
		out.push_back(c);
	}
	return out;
}

// ===========================================================================================
// columns
// ===========================================================================================

int columnValue(const Column& c, int row)
{
	// WordSlice.h:223-229
	u64 mask = row < W - 1 ? ~(kOnes << (row + 1)) : kOnes;
	return c.before + pop(c.vp & mask) - pop(c.vn & mask);
}

namespace {

struct Conf { int rows; bool partial; };
inline bool confLess(Conf a, Conf b) { return a.rows < b.rows || (a.rows == b.rows && !a.partial && b.partial); }   // WordSlice.h:150-153
inline bool confGreater(Conf a, Conf b) { return a.rows > b.rows || (a.rows == b.rows && a.partial && !b.partial); } // WordSlice.h:146-149
inline bool confEq(Conf a, Conf b) { return a.rows == b.rows && a.partial == b.partial; }

// position, in the row-interleaved sequence (vp bit of row i, then "not vn" bit of row i),
// of the rank-th set unit restricted to rows [lo, hi); mirrors BitPosition over the Morton
// words (WordSlice.h:45-97, 99-130, 479-488).  Returns 128+excess when rank runs past the end.
int interleavedRank(u64 vp, u64 vn, int lo, int hi, int rank)
{
	int seen = 0;
	for (int i = lo; i < hi; i++)
	{
		if (bitAt(vp, i)) { if (seen == rank) return 2 * i; seen++; }
		if (!bitAt(vn, i)) { if (seen == rank) return 2 * i + 1; seen++; }
	}
	return 128 + (rank - seen);
}

// WordSlice.h:423-510
Conf mergedConfirmation(Column l, Column r)
{
	Conf lc{l.rows, l.partial}, rc{r.rows, r.partial};
	if (confEq(lc, rc)) return lc;
	if (confGreater(rc, lc)) { std::swap(l, r); std::swap(lc, rc); }
	u64 low = rc.rows >= W ? kOnes : ~(kOnes << rc.rows);
	int ls = l.before + pop(l.vp & low) - pop(l.vn & low);
	int rs = r.before + pop(r.vp & low) - pop(r.vn & low);
	if (rc.rows == lc.rows)
	{
		// right fully confirmed to rows, left has one tentative row more
		rs -= 1;
		if (!bitAt(l.vp, lc.rows & 63)) ls -= 1;
		return {lc.rows, ls <= rs};
	}
	ls += bitAt(l.vp, rc.rows) - bitAt(l.vn, rc.rows);
	if (!(rc.partial && bitAt(r.vp, rc.rows))) rs -= 1;
	if (ls == rs + 1) return {rc.rows, true};
	if (ls > rs + 1) return rc;
	if (lc.rows > rc.rows + 1)
	{
		GAO_CHECK(ls <= rs);
		int lo = rc.rows + 1, hi = lc.rows;
		int p = interleavedRank(l.vp, l.vn, lo, hi, rs - ls);
		if (p / 2 < lc.rows)
		{
			int q = interleavedRank(l.vp, l.vn, lo, hi, rs - ls + 1);
			return {p / 2, q / 2 > p / 2};
		}
		u64 span = (hi >= W ? kOnes : ~(kOnes << hi)) & (kOnes << lo);
		ls += pop(l.vp & span) - pop(l.vn & span);
		rs -= lc.rows - rc.rows - 1;
	}
	if (!lc.partial) return lc;
	rs -= 1;
	if (bitAt(l.vp, lc.rows & 63))
	{
		if (ls <= rs) return lc;
	}
	else
	{
		return lc;
	}
	return {lc.rows, false};
}

}  // namespace

Column mergeColumns(Column a, Column b)
{
	// WordSlice.h:361-421: the result is the cell-wise minimum of the two columns over rows
	// j-1 .. j+63, re-encoded as vertical deltas.  The reference derives the same thing with
	// SWAR prefix sums (differenceMasks, :512-615); here it is computed row by row.
	if (a.before > b.before) std::swap(a, b);
	Conf conf = mergedConfirmation(a, b);
	Column out;
	out.before = a.before;                 // min of the two (a has the smaller one)
	int sa = a.before, sb = b.before, prev = a.before;
	for (int r = 0; r < W; r++)
	{
		sa += bitAt(a.vp, r) - bitAt(a.vn, r);
		sb += bitAt(b.vp, r) - bitAt(b.vn, r);
		int m = std::min(sa, sb);
		if (m == prev + 1) out.vp |= 1ull << r;
		else if (m == prev - 1) out.vn |= 1ull << r;
		prev = m;
	}
	out.end = std::min(a.end, b.end);
	GAO_CHECK(out.end == prev);
	if (a.before < b.before) out.beforeExists = a.beforeExists;          // :398-409
	else out.beforeExists = a.beforeExists || b.beforeExists;
	out.rows = conf.rows;
	out.partial = conf.partial;
	out.endExists = true;                  // default-constructed result (:183)
	return out;
}

Column stepColumn(u64 eq, Column c, bool upIn, bool upLeftIn, bool diagIn, bool prevRowEq, const Column& above, int lastRowMin)
{
	// GraphAligner.h:1349-1427 -- one Myers/Hyyro step with graph-aware horizontal input.
	const int oldBefore = c.before;
	const unsigned cr = (unsigned)c.rows;
	const u64 atConfirmed = 1ull << (cr & 63);          // x86 shift semantics for rows == 64
	const u64 belowConfirmed = 1ull << ((cr - 1) & 63);
	bool oneMore = false;
	if (!c.beforeExists) eq &= ~1ull;
	c.beforeExists = upIn;
	if (!diagIn) eq &= ~1ull;
	if (!upLeftIn)
	{
		c.before += 1;
	}
	else
	{
		GAO_CHECK(c.before <= above.end);
		int viaDiagonal = above.end - bitAt(above.vp, 63) + bitAt(above.vn, 63) + (prevRowEq ? 0 : 1);
		c.before = std::min(c.before + 1, viaDiagonal);
	}
	const int hin = c.before - oldBefore;
	u64 xv = eq | c.vn;
	if (hin < 0) eq |= 1;
	u64 xh = (((eq & c.vp) + c.vp) ^ c.vp) | eq;
	u64 ph = c.vn | ~(xh | c.vp);
	u64 mh = c.vp & xh;
	int diagDiff = hin;
	if (cr > 0) diagDiff = ((ph & belowConfirmed) ? 1 : 0) - ((mh & belowConfirmed) ? 1 : 0);
	if (cr > 0 && (mh & belowConfirmed)) oneMore = true;
	else if (cr == 0 && hin == -1) oneMore = true;
	if (ph >> 63) c.end += 1;
	else if (mh >> 63) c.end -= 1;
	if (c.partial && (~ph & atConfirmed)) oneMore = true;
	ph <<= 1;
	mh <<= 1;
	if (hin < 0) mh |= 1; else if (hin > 0) ph |= 1;
	c.vp = mh | ~(xv | ph);
	c.vn = ph & xv;
	diagDiff += ((c.vp & atConfirmed) ? 1 : 0) - ((c.vn & atConfirmed) ? 1 : 0);
	if (diagDiff <= 0) oneMore = true;
	else if (c.vn & atConfirmed) oneMore = true;
	if (oneMore)
	{
		if (c.rows + 1 <= W) c.rows += 1;
		c.partial = false;
	}
	else if (!c.partial && c.rows < W)
	{
		c.partial = true;
	}
	GAO_CHECK(c.end == c.before + pop(c.vp) - pop(c.vn));                       // :1421
	GAO_CHECK(c.rows < W || c.before >= lastRowMin);                            // :1422
	GAO_CHECK(c.rows < W || c.end >= lastRowMin);                               // :1423
	return c;
}

// WordSlice::setValue (WordSlice.h:231-337): one cell of a column written by the sparse method.  A column is filled from
// row 0 downwards; rows below the last one written are extrapolated with +1 per row.
void setCell(Column& c, int row, int value)
{
	c.written |= 1ull << row;
	if (!c.partial)
	{
		// first cell of the column: rows above it fall by one per row towards it (VN), rows below rise (:236-252)
		c.partial = true;
		c.before = value + row + 1;
		if (row < W - 1) { c.vn = ~(kOnes << (row + 1)); c.vp = kOnes << (row + 1); }
		else { c.vn = kOnes; c.vp = 0; }
		c.rows = row;
		c.end = value + W - row - 1;
		return;
	}
	GAO_CHECK(c.rows < row);
	if (c.rows == row - 1)
	{
		// the row right below the last one written (:254-280)
		const int old = c.end - (W - c.rows - 1);
		GAO_CHECK(old == columnValue(c, c.rows));
		GAO_CHECK(value >= old - 1);
		GAO_CHECK(value <= old + 1);
		const u64 m = 1ull << row;
		if (value == old - 1) { c.vn |= m; c.vp &= ~m; c.end -= 2; }
		else if (value == old) { c.vn &= ~m; c.vp &= ~m; c.end -= 1; }
		else { c.vp |= m; c.vn &= ~m; }
		c.rows = row;
		return;
	}
	// a gap: the rows between rise by one each, then every row down to `row` is capped by the run going up from the new cell (:281-336)
	int sc[W];
	sc[0] = c.before + bitAt(c.vp, 0) - bitAt(c.vn, 0);
	for (int i = 1; i <= c.rows; i++) sc[i] = sc[i - 1] + bitAt(c.vp, i) - bitAt(c.vn, i);
	for (int i = c.rows + 1; i <= row; i++) sc[i] = sc[i - 1] + 1;
	for (int i = 0; i <= row; i++) sc[i] = std::min(sc[i], value + row - i);
	GAO_CHECK(sc[0] >= c.before - 1);
	GAO_CHECK(sc[0] <= c.before + 1);
	auto put = [&](int i, int delta) {
		const u64 m = 1ull << i;
		if (delta == -1) { c.vp &= ~m; c.vn |= m; }
		else if (delta == 0) { c.vp &= ~m; c.vn &= ~m; }
		else { c.vp |= m; c.vn &= ~m; }
	};
	put(0, sc[0] - c.before);
	for (int i = 1; i <= row; i++)
	{
		GAO_CHECK(sc[i] >= sc[i - 1] - 1);
		GAO_CHECK(sc[i] <= sc[i - 1] + 1);
		put(i, sc[i] - sc[i - 1]);
	}
	c.end = sc[row] + W - 1 - row;
	c.rows = row;
}

// ===========================================================================================
// HMM
// ===========================================================================================

namespace {
struct HmmTables {
	double cMis, cMat, fMis, fMat, f2c, f2f, c2f, c2c;
	double logFact[65];
	HmmTables()
	{
		// AlignmentCorrectnessEstimation.cpp:6-28
		cMis = log(0.2); cMat = log(1.0 - 0.2); fMis = log(0.5); fMat = log(1.0 - 0.5);
		f2c = log(0.00001); f2f = log(1.0 - 0.00001);
		c2f = log(0.000000000000001); c2c = log(1.0 - 0.000000000000001);
		logFact[0] = 0;
		for (int i = 1; i <= 64; i++) logFact[i] = logFact[i - 1] + log(i);
	}
};
const HmmTables& hmmTables() { static const HmmTables t; return t; }
}  // namespace

Hmm::Hmm() : correct(log(0.8)), wrong(log(0.2)) {}                            // :30-36

Hmm Hmm::next(int mismatches, int rowSize) const
{
	// AlignmentCorrectnessEstimation.cpp:71-89
	const HmmTables& t = hmmTables();
	GAO_CHECK(rowSize == 64 || rowSize == 1);
	GAO_CHECK(mismatches >= 0);
	GAO_CHECK(mismatches <= rowSize);
	Hmm r;
	r.correctFromCorrect = correct + t.c2c >= wrong + t.f2c;
	r.falseFromCorrect = correct + t.c2f >= wrong + t.f2f;
	double nc = std::max(correct + t.c2c, wrong + t.f2c);
	double nf = std::max(correct + t.c2f, wrong + t.f2f);
	double choose = t.logFact[rowSize] - t.logFact[mismatches] - t.logFact[rowSize - mismatches];
	double cm = choose + mismatches * t.cMis + (rowSize - mismatches) * t.cMat;
	double fm = choose + mismatches * t.fMis + (rowSize - mismatches) * t.fMat;
	nc += cm;
	nf += fm;
	r.correct = nc;
	r.wrong = nf;
	return r;
}

// ===========================================================================================
// slice storage (NodeSlice.h)
// ===========================================================================================

namespace {

struct Span { size_t lo = 0, hi = 0; int minScore = 0; };                    // NodeSlice::MapItem (:430)

struct EndCell { uint16_t plus; uint8_t bits; };                             // TinySlice (:26-31)
struct FullCell { u64 vp, vn; uint16_t plus; bool endExists; };              // SmallSlice (:15-25)

class Store
{
public:
	Store() {}
	explicit Store(std::vector<Span>* denseMap) : dense(denseMap) {}

	// --- node map -----------------------------------------------------------------------
	void addNode(size_t node, size_t len)                                    // NodeSlice.h:584-599
	{
		size_t lo = size();
		if (dense)
		{
			GAO_CHECK(node < dense->size());
			GAO_CHECK((*dense)[node].lo == (*dense)[node].hi);
			(*dense)[node] = Span{lo, lo + len, 0};
			active.push_back(node);
		}
		else
		{
			GAO_CHECK(sparse.find(node) == sparse.end());
			sparse[node] = Span{lo, lo + len, 0};
		}
		GAO_CHECK(mode == 0);
		live.resize(lo + len);
	}
	bool has(size_t node) const                                              // :630-641
	{
		if (dense) return (*dense)[node].lo != (*dense)[node].hi;
		return sparse.find(node) != sparse.end();
	}
	Span span(size_t node) const                                             // :600-629
	{
		if (dense)
		{
			GAO_CHECK((*dense)[node].lo != (*dense)[node].hi);
			return (*dense)[node];
		}
		auto it = sparse.find(node);
		GAO_CHECK(it != sparse.end());
		return it->second;
	}
	void setMin(size_t node, int score)                                      // :657-668
	{
		if (dense) (*dense)[node].minScore = score; else sparse[node].minScore = score;
	}
	size_t nodeCount() const { return dense ? active.size() : sparse.size(); }
	void releaseDense()                                                      // clearVectorMap (:571-579)
	{
		GAO_CHECK(dense != nullptr);
		for (size_t n : active) (*dense)[n] = Span{};
		active.clear();
	}
	// iteration in the container's own order (:680-723): dense -> insertion order,
	// detached -> std::unordered_map order
	template <typename F> void forEach(F f) const
	{
		if (dense) { for (size_t n : active) f(n, (*dense)[n]); }
		else { for (const auto& kv : sparse) f(kv.first, kv.second); }
	}

	// --- cells --------------------------------------------------------------------------
	size_t size() const { return mode == 0 ? live.size() : mode == 1 ? full.size() : ends.size(); }
	Column& at(size_t i) { GAO_CHECK(mode == 0); return live[i]; }           // WordContainer::operator[] (:295-299)
	Column get(size_t i) const                                               // const operator[] (:300-326)
	{
		if (mode == 1)
		{
			Column c;
			c.vp = full[i].vp; c.vn = full[i].vn;
			c.end = 0;                                                       // the reference leaves scoreEnd 0 here (:304)
			c.before = minBefore + full[i].plus;
			c.rows = 64; c.partial = false; c.beforeExists = false;
			c.endExists = full[i].endExists;
			return c;
		}
		if (mode == 2)
		{
			bool p = ends[i].bits & 1, n = ends[i].bits & 2;
			Column c;
			c.vp = (u64)p << 63; c.vn = (u64)n << 63;
			c.end = minEnd + ends[i].plus;
			c.before = c.end - (p ? 1 : 0) + (n ? 1 : 0);
			c.rows = 64; c.partial = false; c.beforeExists = false;
			c.endExists = ends[i].bits & 4;
			return c;
		}
		return live[i];
	}

	// --- freezing -------------------------------------------------------------------------
	Store frozenEnds() const                                                 // NodeSlice.h:724-740 + 353-376
	{
		Store r;
		if (mode == 2) { r.mode = 2; r.ends = ends; r.minEnd = minEnd; }
		else
		{
			GAO_CHECK(mode == 0);
			r.mode = 2;
			r.ends.resize(live.size());
			r.minEnd = live[0].end;
			for (size_t i = 1; i < live.size(); i++) r.minEnd = std::min(r.minEnd, live[i].end);
			for (size_t i = 0; i < live.size(); i++)
			{
				uint8_t b = 0;
				b |= (uint8_t)(live[i].vp >> 63);
				b |= (uint8_t)((live[i].vn >> 62) & 2);
				b |= (uint8_t)(live[i].endExists << 2);
				GAO_CHECK(live[i].end >= r.minEnd);
				GAO_CHECK(live[i].end - r.minEnd < 65535);
				r.ends[i] = EndCell{(uint16_t)(live[i].end - r.minEnd), b};
			}
		}
		copyMapInto(r);
		return r;
	}
	Store frozenFull() const                                                 // NodeSlice.h:741-757 + 327-352
	{
		Store r;
		if (mode == 1) { r.mode = 1; r.full = full; r.minBefore = minBefore; }
		else
		{
			GAO_CHECK(mode == 0);
			r.mode = 1;
			r.full.resize(live.size());
			r.minBefore = live[0].before;
			for (size_t i = 1; i < live.size(); i++) r.minBefore = std::min(r.minBefore, live[i].before);
			for (size_t i = 0; i < live.size(); i++)
			{
				GAO_CHECK(live[i].before >= r.minBefore);
				GAO_CHECK(live[i].before - r.minBefore < 65535);
				r.full[i] = FullCell{live[i].vp, live[i].vn, (uint16_t)(live[i].before - r.minBefore), live[i].endExists};
			}
		}
		copyMapInto(r);
		return r;
	}

private:
	void copyMapInto(Store& r) const
	{
		// a detached map is (re)built by inserting the active nodes one by one, in band order,
		// into an empty std::unordered_map (:728-738) -- this fixes its iteration order
		if (dense) { for (size_t n : active) r.sparse[n] = (*dense)[n]; }
		else r.sparse = sparse;
	}

	int mode = 0;
	std::vector<Column> live;
	std::vector<FullCell> full;
	std::vector<EndCell> ends;
	int minEnd = 0, minBefore = 0;
	std::vector<Span>* dense = nullptr;
	std::vector<size_t> active;
	std::unordered_map<size_t, Span> sparse;
};

struct Slice                                                                 // DPSlice (GraphAligner.h:105-166)
{
	Slice() {}
	explicit Slice(std::vector<Span>* denseMap) : cells(denseMap) {}
	int minScore = std::numeric_limits<int>::min();
	std::vector<size_t> minIndex;
	Store cells;
	std::vector<size_t> nodes;
	Hmm hmm;
	size_t j = std::numeric_limits<size_t>::max();
	size_t cellsProcessed = 0;
	size_t numCells = 0;
	bool sparse = false;                                                     // (bookkeeping for tests: which method filled the slice)
	size_t estimatedMemory() const { return numCells * 4 + cells.nodeCount() * (sizeof(size_t) * 3 + sizeof(int)); }   // :136-139
	Slice withCells(Store s) const
	{
		Slice r;
		r.cells = std::move(s);
		r.minScore = minScore; r.minIndex = minIndex; r.nodes = nodes; r.hmm = hmm; r.j = j;
		r.cellsProcessed = cellsProcessed; r.numCells = numCells;
		return r;
	}
	Slice frozenEnds() const { return withCells(cells.frozenEnds()); }       // :140-152
	Slice frozenFull() const { return withCells(cells.frozenFull()); }       // :153-165
};

typedef std::pair<size_t, size_t> Pos;                                       // (column, row)

// BacktraceOverride (GraphAligner.h:167-354): the predecessor of every cell that can be reached backwards from an existing
// end cell of a window of slices, worked out while the window's full slices are still in memory
struct Override
{
	struct Item { bool end = false; bool sameRow = false; size_t previous = 0; Pos pos{0, 0}; };
	size_t startj = 0, endj = 0;
	std::vector<std::vector<Item>> items;                                    // per row of the window
};

struct Table                                                                 // DPTable (:355-367)
{
	std::vector<Slice> slices;
	size_t samplingFrequency = 0;
	std::vector<size_t> bandwidthPerSlice;
	std::vector<Hmm> correctness;
	std::vector<Override> overrides;
};

struct NodeCalc { int minScore; std::vector<size_t> minIndex; size_t cellsProcessed; };

struct Prio                                                                  // NodeWithPriority (:1094-1108)
{
	Prio(size_t n, int p) : node(n), priority(p) {}
	bool operator>(const Prio& o) const { return priority > o.priority; }
	bool operator<(const Prio& o) const { return priority < o.priority; }
	size_t node; int priority;
};
typedef std::priority_queue<Prio, std::vector<Prio>, std::greater<Prio>> MinQueue;

// LIFO work list with membership flags (UniqueQueue.h:6-69)
struct WorkStack
{
	explicit WorkStack(size_t n) : member(n, false) {}
	void push(size_t v) { if (member[v]) return; member[v] = true; items.push_back(v); }
	size_t top() const { return items.back(); }
	void pop() { member[items.back()] = false; items.pop_back(); }
	size_t size() const { return items.size(); }
	std::vector<size_t> items;
	std::vector<bool> member;
};

// ===========================================================================================
// the engine
// ===========================================================================================

class Engine
{
public:
	Engine(const Graph& graph, int bw, int rampBw, std::vector<SliceRecord>* rec) : g(graph), initialBandwidth(bw), rampBandwidth(rampBw), record(rec) {}

	AlignResult align(const std::string& seqId, const std::string& sequence, const std::vector<Seed>& seeds);

private:
	const Graph& g;
	const int initialBandwidth, rampBandwidth;
	std::vector<SliceRecord>* record;
	int recordDirection = 0;
	bool countSparse = true;                                                 // false while slices are recomputed for the traceback
	int lastRowMin = 0;                                                      // debugLastRowMinScore (:54-56)
	size_t statColumns = 0, statSlices = 0, statSparse = 0, statOverrides = 0, statOverrideTraces = 0;

	struct Split { size_t splitIndex = 0; Table forward, backward; size_t estimated() const { return (forward.bandwidthPerSlice.size() + backward.bandwidthPerSlice.size()) * W; } };
	typedef std::pair<int, std::vector<Pos>> Trace;

	// --- cell access ---------------------------------------------------------------------
	int cellValue(const Slice& s, size_t row, size_t column) const           // getValue (:2019-2027)
	{
		size_t n = g.nodeOf(column);
		Span sp = s.cells.span(n);
		return columnValue(s.cells.get(sp.lo + (column - g.nodeBegin(n))), (int)(row % W));
	}
	int cellValueOr(const Slice& s, size_t row, size_t column, int dflt) const   // getValueOrMax (:2008-2017)
	{
		size_t n = g.nodeOf(column);
		if (!s.cells.has(n)) return dflt;
		Span sp = s.cells.span(n);
		return columnValue(s.cells.get(sp.lo + (column - g.nodeBegin(n))), (int)(row % W));
	}

	Pos pickPredecessor(const std::string& seq, const Slice& slice, Pos pos, const Slice& before) const;
	std::vector<size_t> projectBand(int minScore, const Slice& previous, int bandwidth) const;
	std::vector<std::vector<size_t>> components(const std::vector<size_t>& order, const std::vector<bool>& inBand) const;
	void zeroRow(Store& cur, const Store& prev, const std::vector<bool>& curBand, const std::vector<bool>& prevBand,
	             const std::vector<size_t>& comp, size_t compIndex, const std::vector<size_t>& compOf) const;
	Column nodeStartColumn(u64 eq, size_t node, const Store& prev, const Store& cur, const std::vector<bool>& curBand,
	                       const std::vector<bool>& prevBand, bool prevRowEq) const;
	NodeCalc fillNode(size_t node, size_t j, const std::string& seq, const u64 eqOf[4], Store& cur, const Store& prev,
	                  const std::vector<bool>& curBand, const std::vector<bool>& prevBand) const;
	NodeCalc fillSlice(const std::string& seq, size_t j, Store& cur, const Store& prev, const std::vector<size_t>& order,
	                   const std::vector<bool>& curBand, const std::vector<bool>& prevBand, std::vector<size_t>& compOf, WorkStack& work) const;
	Slice extendAndFill(const std::string& seq, const Slice& previous, const std::vector<bool>& prevBand, std::vector<bool>& curBand,
	                    std::vector<size_t>& compOf, WorkStack& work, std::vector<bool>& processed, std::vector<Span>& dense, int bandwidth);
	NodeCalc fillSliceSparse(const std::string& seq, size_t startj, Store& cur, const Slice& previous, std::vector<bool>& processed, int bandwidth) const;
	void finishSparseSlice(Slice& s, std::vector<bool>& curBand, int uninitialized, int bandwidth) const;
	Override makeOverride(const std::string& seq, const Slice& previous, const std::vector<Slice>& window) const;
	std::vector<Pos> overrideBacktrace(const Override& o, Pos start) const;
	Table firstPass(const std::string& seq, const Slice& initial, size_t numSlices, size_t samplingFrequency, std::vector<Span>& dense);
	std::vector<Slice> recompute(const std::string& seq, size_t overrideLastJ, const Table& table, size_t startIndex, std::vector<Span>& dense);
	void trimWrongEnd(Table& t) const;
	Slice seedSlice(size_t node) const;
	Split splitAlign(const std::string& sequence, int bigraphNode, bool backwards, size_t pos, std::vector<Span>& dense);
	Trace traceTable(const std::string& seq, const Table& table, std::vector<Span>& dense);
	std::vector<Pos> traceInSlice(const std::string& seq, const Slice& s, Pos pos) const;
	std::vector<Pos> traceBoundary(const std::string& seq, const Slice& after, const Slice& before, size_t column) const;
	std::vector<Pos> traceInner(const std::string& seq, const std::vector<Slice>& part, Pos pos) const;
	std::pair<Trace, Trace> piecewise(const Split& split, const std::string& sequence, std::vector<Span>& dense);
	void noteTried(std::vector<std::tuple<size_t, size_t, size_t>>& tried, const std::pair<Trace, Trace>& tr) const;
	std::vector<TraceItem> traceItems(const std::string& seq, const std::vector<Pos>& bw, const std::vector<Pos>& fw) const;
	std::vector<TraceItem> traceItemsInner(const std::string& seq, const std::vector<Pos>& tr) const;

	struct Partial { bool failed; int32_t score; std::vector<Mapping> mappings; };
	Partial toMappings(const std::string& sequence, int score, const std::vector<Pos>& trace) const;
	Partial mergePartials(const Partial& first, const Partial& second) const;
	static u64 eqFor(const u64 eqOf[4], char graphChar)
	{
		// EqVector::getEq (:77-99)
		switch (graphChar) { case 'A': return eqOf[0]; case 'T': return eqOf[1]; case 'C': return eqOf[2]; case 'G': return eqOf[3]; }
		fail(ASSERTION, "Eq for non-ACGT graph char");
	}
};

// ------------------------------------------------------------------------------------------
// backtrace predecessor (GraphAligner.h:493-591)
// ------------------------------------------------------------------------------------------
Pos Engine::pickPredecessor(const std::string& seq, const Slice& slice, Pos pos, const Slice& before) const
{
	const size_t w = pos.first, row = pos.second;
	GAO_CHECK(row >= slice.j);
	GAO_CHECK(row < slice.j + W);
	const int big = (int)seq.size();
	size_t node = g.nodeOf(w);
	GAO_CHECK(slice.cells.has(node));
	const int here = cellValue(slice, row - slice.j, w);
	if (row == 0 && before.cells.has(node) && (here == 0 || here == 1)) return Pos{w, row - 1};
	auto diagonalOf = [&](size_t u) {
		return row == slice.j ? cellValueOr(before, W - 1, u, big) : cellValueOr(slice, row - 1 - slice.j, u, big);
	};
	auto tryFrom = [&](size_t u, Pos& out) -> bool {
		int horizontal = cellValueOr(slice, row - slice.j, u, big);
		GAO_CHECK(horizontal >= here - 1);
		if (horizontal == here - 1) { out = Pos{u, row}; return true; }
		int diagonal = diagonalOf(u);
		if (charMatch(seq[row], g.base(w)))
		{
			GAO_CHECK(diagonal >= here);
			if (diagonal == here) { out = Pos{u, row - 1}; return true; }
		}
		else
		{
			GAO_CHECK(diagonal >= here - 1);
			if (diagonal == here - 1) { out = Pos{u, row - 1}; return true; }
		}
		return false;
	};
	Pos out;
	if (w == g.nodeBegin(node))
	{
		for (size_t nb : g.in[node]) if (tryFrom(g.nodeEnd(nb) - 1, out)) return out;
	}
	else
	{
		if (tryFrom(w - 1, out)) return out;
	}
	int up;
	if (row == slice.j)
	{
		GAO_CHECK(before.j + W == slice.j);
		up = cellValueOr(before, W - 1, w, big);
	}
	else up = cellValueOr(slice, row - 1 - slice.j, w, big);
	GAO_CHECK(up >= here - 1);
	if (up == here - 1) return Pos{w, row - 1};
	fail(ASSERTION, "no backtrace predecessor");
}

// ------------------------------------------------------------------------------------------
// band selection (GraphAligner.h:1110-1159)
// ------------------------------------------------------------------------------------------
std::vector<size_t> Engine::projectBand(int minScore, const Slice& previous, int bandwidth) const
{
	const int expand = bandwidth + W;
	std::unordered_map<size_t, size_t> dist;
	std::vector<size_t> band;
	MinQueue queue;
	size_t width = 0;
	bool full = false;
	previous.cells.forEach([&](size_t node, const Span& sp) {
		if (full) return;
		if (sp.minScore <= minScore + bandwidth)
		{
			dist[node] = 0;
			band.push_back(node);
			width += g.nodeLen(node);
			if (width >= kCutoff) { full = true; return; }
			int endScore = previous.cells.get(sp.hi - 1).end;
			GAO_CHECK(endScore >= minScore);
			if (endScore > minScore + expand) return;
			for (size_t nb : g.out[node]) queue.emplace(nb, endScore - minScore + 1);
		}
	});
	if (full) return band;
	GAO_CHECK(dist.size() > 0);
	while (queue.size() > 0)
	{
		Prio top = queue.top();
		if (top.priority > expand) break;
		queue.pop();
		auto it = dist.find(top.node);
		if (it != dist.end() && it->second <= (size_t)top.priority) continue;
		width += g.nodeLen(top.node);
		dist[top.node] = top.priority;
		band.push_back(top.node);
		if (width >= kCutoff) return band;
		int len = (int)g.nodeLen(top.node);
		for (size_t nb : g.out[top.node]) queue.emplace(nb, top.priority + len);
	}
	return band;
}

// ------------------------------------------------------------------------------------------
// strongly connected components, Tarjan, emission order preserved (GraphAligner.h:1751-1901)
// ------------------------------------------------------------------------------------------
std::vector<std::vector<size_t>> Engine::components(const std::vector<size_t>& order, const std::vector<bool>& inBand) const
{
	std::vector<std::vector<size_t>> result;
	std::unordered_map<size_t, size_t> index, low;
	std::unordered_set<size_t> onStack;
	std::vector<size_t> stack;
	size_t counter = 0;
	struct Frame { size_t node; size_t next; };
	for (size_t root : order)
	{
		GAO_CHECK(inBand[root]);
		if (index.count(root)) continue;
		std::vector<Frame> frames;
		auto open = [&](size_t v) {
			index[v] = counter; low[v] = counter; counter++;
			stack.push_back(v); onStack.insert(v);
			frames.push_back(Frame{v, 0});
		};
		open(root);
		while (!frames.empty())
		{
			Frame& f = frames.back();
			const auto& outs = g.out[f.node];
			if (f.next < outs.size())
			{
				size_t nb = outs[f.next];
				if (!inBand[nb]) { f.next++; continue; }
				if (!index.count(nb))
				{
					// descend; on return the child's low-link is folded in (state 1 of the reference)
					open(nb);
					continue;
				}
				if (onStack.count(nb)) low[f.node] = std::min(low[f.node], index[nb]);
				f.next++;
				continue;
			}
			size_t v = f.node;
			frames.pop_back();
			if (low[v] == index[v])
			{
				result.emplace_back();
				size_t back;
				do
				{
					back = stack.back();
					result.back().push_back(back);
					onStack.erase(back);
					stack.pop_back();
				} while (back != v);
			}
			if (!frames.empty())
			{
				Frame& parent = frames.back();
				low[parent.node] = std::min(low[parent.node], low[v]);
				parent.next++;
			}
		}
	}
	GAO_CHECK(stack.empty());
	GAO_CHECK(index.size() == order.size());
	return result;
}

// ------------------------------------------------------------------------------------------
// exact scores of the virtual row j-1 for one component (GraphAligner.h:1903-1995)
// ------------------------------------------------------------------------------------------
void Engine::zeroRow(Store& cur, const Store& prev, const std::vector<bool>& curBand, const std::vector<bool>& prevBand,
                     const std::vector<size_t>& comp, size_t compIndex, const std::vector<size_t>& compOf) const
{
	const int INF = std::numeric_limits<int>::max();
	MinQueue queue;
	for (size_t node : comp)
	{
		GAO_CHECK(curBand[node]);
		Span me = cur.span(node);
		size_t len = me.hi - me.lo;
		for (size_t i = 0; i < len; i++) cur.at(me.lo + i).before = INF;
		Span old = prevBand[node] ? prev.span(node) : Span{};
		if (prevBand[node]) cur.at(me.lo).before = prev.get(old.lo).end;
		for (size_t nb : g.in[node])
		{
			if (!curBand[nb] && !prevBand[nb]) continue;
			if (compOf[nb] == compIndex) continue;
			if (curBand[nb])
			{
				Span s = cur.span(nb);
				GAO_CHECK(cur.get(s.hi - 1).rows == W);
				cur.at(me.lo).before = std::min(cur.at(me.lo).before, cur.get(s.hi - 1).before + 1);
			}
			if (prevBand[nb])
			{
				Span s = prev.span(nb);
				cur.at(me.lo).before = std::min(cur.at(me.lo).before, prev.get(s.hi - 1).end + 1);
			}
		}
		if (cur.at(me.lo).before == INF) continue;
		for (size_t i = 1; i < len; i++)
		{
			int v = cur.at(me.lo + i - 1).before + 1;
			if (prevBand[node]) v = std::min(v, prev.get(old.lo + i).end);
			cur.at(me.lo + i).before = v;
		}
		for (size_t nb : g.out[node])
		{
			if (compOf[nb] != compIndex) continue;
			queue.emplace(nb, cur.at(me.hi - 1).before + 1);
		}
	}
	while (queue.size() > 0)
	{
		Prio top = queue.top();
		queue.pop();
		int score = top.priority;
		Span s = cur.span(top.node);
		bool reachedEnd = true;
		for (size_t i = s.lo; i < s.hi; i++)
		{
			if (cur.at(i).before <= score) { reachedEnd = false; break; }
			cur.at(i).before = score;
			score++;
		}
		if (reachedEnd)
		{
			for (size_t nb : g.out[top.node])
			{
				if (compOf[nb] != compIndex) continue;
				queue.emplace(nb, score);
			}
		}
	}
	for (size_t node : comp)
	{
		Span me = cur.span(node);
		Span old = prevBand[node] ? prev.span(node) : Span{};
		for (size_t i = 0; i < me.hi - me.lo; i++)
		{
			int b = cur.at(me.lo + i).before;
			GAO_CHECK(b != INF);
			Column c;
			c.vp = kOnes; c.vn = 0; c.end = b + W; c.before = b; c.rows = 0; c.partial = false;
			c.beforeExists = false;
			if (prevBand[node])
			{
				Column o = prev.get(old.lo + i);
				c.beforeExists = o.end == b && o.endExists;
			}
			c.endExists = true;
			cur.at(me.lo + i) = c;
		}
	}
}

// ------------------------------------------------------------------------------------------
// first column of a node from its in-neighbours (GraphAligner.h:1270-1315)
// ------------------------------------------------------------------------------------------
Column Engine::nodeStartColumn(u64 eq, size_t node, const Store& prev, const Store& cur, const std::vector<bool>& curBand,
                               const std::vector<bool>& prevBand, bool prevRowEq) const
{
	const Column mine = cur.get(cur.span(node).lo);
	Column result;
	bool any = false;
	for (size_t nb : g.in[node])
	{
		if (!curBand[nb] && !prevBand[nb]) continue;
		u64 eqHere = eq;
		Column left, above;
		bool haveAbove = false;
		if (prevBand[nb])
		{
			above = prev.get(prev.span(nb).hi - 1);
			haveAbove = true;
		}
		if (curBand[nb])
		{
			left = cur.get(cur.span(nb).hi - 1);
		}
		else
		{
			// neighbour only in the previous band: a vertical source column whose only possible
			// match is the diagonal into row j (:1294-1301)
			int s = prev.get(prev.span(nb).hi - 1).end;
			left = Column{};
			left.vp = kOnes; left.vn = 0; left.end = s + W; left.before = s; left.rows = W;
			left.beforeExists = true;
			eqHere &= 1;
		}
		Column here = stepColumn(eqHere, left, mine.beforeExists, mine.beforeExists && haveAbove, haveAbove, prevRowEq, above, lastRowMin);
		if (!any) { result = here; any = true; }
		else result = mergeColumns(result, here);
	}
	GAO_CHECK(any);
	return result;
}

// ------------------------------------------------------------------------------------------
// all columns of one node for one slice (GraphAligner.h:1457-1573)
// ------------------------------------------------------------------------------------------
NodeCalc Engine::fillNode(size_t node, size_t j, const std::string& seq, const u64 eqOf[4], Store& cur, const Store& prev,
                          const std::vector<bool>& curBand, const std::vector<bool>& prevBand) const
{
	NodeCalc res;
	res.minScore = std::numeric_limits<int>::max();
	res.cellsProcessed = 0;
	const Span me = cur.span(node);
	const size_t len = me.hi - me.lo;
	GAO_CHECK(len == g.nodeLen(node));
	const size_t first = g.nodeBegin(node);
	const bool inPrev = prevBand[node];
	const Span old = inPrev ? prev.span(node) : Span{};
	auto above = [&](size_t i) { return inPrev ? prev.get(old.lo + i) : cur.get(me.lo + i); };
	auto prevRowEq = [&](size_t i) { return (j == 0 && inPrev) || (j > 0 && g.base(first + i) == seq[j - 1]); };
	auto note = [&](size_t i) {
		const Column& c = cur.at(me.lo + i);
		if (c.rows == W && c.end < res.minScore) { res.minScore = c.end; res.minIndex.clear(); }
		if (c.rows == W && c.end == res.minScore) res.minIndex.push_back(first + i);
	};
	auto verticalEntry = [&](size_t i) {
		// re-entry from the cell above when it is better than what came from the left (:1504-1509, 1541-1546)
		if (!inPrev) return;
		Column o = prev.get(old.lo + i);
		if (cur.at(me.lo + i).before > o.end)
		{
			Column src;
			src.vp = kOnes; src.vn = 0; src.end = o.end + W; src.before = o.end; src.rows = W;
			src.beforeExists = o.endExists;
			cur.at(me.lo + i) = mergeColumns(cur.at(me.lo + i), src);
		}
	};
	auto sanity = [&](size_t i) {
		// assertSliceCorrectness (:1437-1455)
		const Column& c = cur.at(me.lo + i);
		GAO_CHECK(c.end == c.before + pop(c.vp) - pop(c.vn));
		GAO_CHECK(c.before >= 0);
		GAO_CHECK(c.end >= 0);
		GAO_CHECK((c.vp & c.vn) == 0);
		GAO_CHECK(!inPrev || c.before <= above(i).end);
		GAO_CHECK(c.rows < W || c.end >= lastRowMin);
		GAO_CHECK(c.rows < W || c.before >= lastRowMin);
	};

	Conf before0{cur.at(me.lo).rows, cur.at(me.lo).partial};
	if (before0.rows == W) return res;

	bool source = true;
	for (size_t nb : g.in[node]) if (curBand[nb] || prevBand[nb]) { source = false; break; }     // isSource (:1339-1347)
	if (source)
	{
		Column c;
		c.vn = 0; c.rows = W;
		if (j == 0 && inPrev)
		{
			int s = prev.get(old.lo).end;
			u64 firstVp = charMatch(seq[0], g.base(first)) ? 0 : 1;                              // :1327-1331
			c.vp = (kOnes & ~1ull) | firstVp; c.end = s + W - 1 + (int)firstVp; c.before = s; c.beforeExists = true;
		}
		else if (inPrev)
		{
			Column o = prev.get(old.lo);                                                         // :1333-1337
			c.vp = kOnes; c.end = o.end + W; c.before = o.end; c.beforeExists = o.endExists;
		}
		else
		{
			size_t rowv = seq.size();                                                            // :1317-1320
			c.vp = kOnes & ~1ull; c.end = (int)(rowv + W); c.before = (int)(rowv + 1); c.beforeExists = false;
		}
		cur.at(me.lo) = c;
		note(0);
		sanity(0);
	}
	else
	{
		u64 eq = eqFor(eqOf, g.base(first));
		cur.at(me.lo) = nodeStartColumn(eq, node, prev, cur, curBand, prevBand, prevRowEq(0));
		verticalEntry(0);
		note(0);
		sanity(0);
	}
	{
		Conf now{cur.at(me.lo).rows, cur.at(me.lo).partial};
		GAO_CHECK(!confLess(now, before0));
		if (confEq(now, before0)) return res;
	}
	for (size_t w = 1; w < len; w++)
	{
		u64 eq = eqFor(eqOf, g.base(first + w));
		Conf was{cur.at(me.lo + w).rows, cur.at(me.lo + w).partial};
		if (was.rows == W) return res;
		bool e = cur.at(me.lo + w).beforeExists;
		cur.at(me.lo + w) = stepColumn(eq, cur.at(me.lo + w - 1), e, e, cur.at(me.lo + w - 1).beforeExists, prevRowEq(w), above(w - 1), lastRowMin);
		verticalEntry(w);
		GAO_CHECK(inPrev || cur.at(me.lo + w).before == (int)j || cur.at(me.lo + w).before == cur.at(me.lo + w - 1).before + 1);   // :1548
		sanity(w);
		note(w);
		Conf now{cur.at(me.lo + w).rows, cur.at(me.lo + w).partial};
		if (confEq(now, was)) return res;
	}
	res.cellsProcessed = len * W;
	return res;
}

// ------------------------------------------------------------------------------------------
// one 64-row slice over the band (GraphAligner.h:2331-2451)
// ------------------------------------------------------------------------------------------
NodeCalc Engine::fillSlice(const std::string& seq, size_t j, Store& cur, const Store& prev, const std::vector<size_t>& order,
                           const std::vector<bool>& curBand, const std::vector<bool>& prevBand, std::vector<size_t>& compOf, WorkStack& work) const
{
	int best = std::numeric_limits<int>::max();
	std::vector<size_t> bestIndex;
	size_t cellsProcessed = 0;
	u64 eqOf[4] = {0, 0, 0, 0};   // A, T, C, G
	for (int i = 0; i < W && j + i < seq.size(); i++)
	{
		u64 m = 1ull << i;
		if (charMatch(seq[j + i], 'A')) eqOf[0] |= m;
		if (charMatch(seq[j + i], 'C')) eqOf[2] |= m;
		if (charMatch(seq[j + i], 'T')) eqOf[1] |= m;
		if (charMatch(seq[j + i], 'G')) eqOf[3] |= m;
	}
	GAO_CHECK((eqOf[0] | eqOf[1] | eqOf[2] | eqOf[3]) == kOnes);
	auto comps = components(order, curBand);
	for (size_t i = 0; i < comps.size(); i++) for (size_t n : comps[i]) compOf[n] = i;
	for (size_t ci = comps.size(); ci-- > 0;)
	{
		zeroRow(cur, prev, curBand, prevBand, comps[ci], ci, compOf);
		GAO_CHECK(work.size() == 0);
		for (size_t n : comps[ci]) work.push(n);
		while (work.size() > 0)
		{
			size_t n = work.top();
			GAO_CHECK(curBand[n]);
			work.pop();
			Span sp = cur.span(n);
			Column oldEnd = cur.get(sp.hi - 1);
			NodeCalc nc = fillNode(n, j, seq, eqOf, cur, prev, curBand, prevBand);
			cur.setMin(n, nc.minScore);
			Column newEnd = cur.get(sp.hi - 1);
			GAO_CHECK(newEnd.before == oldEnd.before);                                         // :2385
			Conf oc{oldEnd.rows, oldEnd.partial}, ncf{newEnd.rows, newEnd.partial};
			GAO_CHECK(!confLess(ncf, oc));
			if (newEnd.before < (int)seq.size() && confGreater(ncf, oc))
			{
				for (size_t nb : g.out[n])
				{
					if (compOf[nb] != ci) continue;
					if (cur.get(cur.span(nb).lo).rows < W) work.push(nb);
				}
			}
			if (nc.minScore < best) { best = nc.minScore; bestIndex.clear(); }
			if (nc.minScore == best) bestIndex.insert(bestIndex.end(), nc.minIndex.begin(), nc.minIndex.end());
			cellsProcessed += nc.cellsProcessed;
		}
		for (size_t n : comps[ci]) GAO_CHECK(cur.get(cur.span(n).lo).rows == W);               // :2422-2425
	}
	for (size_t i = 0; i < comps.size(); i++) for (size_t n : comps[i]) compOf[n] = std::numeric_limits<size_t>::max();
	return NodeCalc{best, bestIndex, cellsProcessed};
}

// ------------------------------------------------------------------------------------------
// band + fill for one slice (GraphAligner.h:2453-2521)
// ------------------------------------------------------------------------------------------
Slice Engine::extendAndFill(const std::string& seq, const Slice& previous, const std::vector<bool>& prevBand, std::vector<bool>& curBand,
                            std::vector<size_t>& compOf, WorkStack& work, std::vector<bool>& processed, std::vector<Span>& dense, int bandwidth)
{
	{
		Slice s(&dense);
		s.j = previous.j + W;
		s.hmm = previous.hmm;
		s.nodes = projectBand(previous.minScore, previous, bandwidth);
		GAO_CHECK(s.nodes.size() > 0);
		GAO_CHECK(seq.size() >= s.j + W);
		size_t cells = 0;
		for (size_t n : s.nodes) cells += g.nodeLen(n);
		if (cells < kCutoff)                                                                       // :2483
		{
			for (size_t n : s.nodes)
			{
				s.cells.addNode(n, g.nodeLen(n));
				curBand[n] = true;
			}
			NodeCalc r = fillSlice(seq, s.j, s.cells, previous.cells, s.nodes, curBand, prevBand, compOf, work);
			s.cellsProcessed = r.cellsProcessed;
			s.minIndex = r.minIndex;
			s.minScore = r.minScore;
			GAO_CHECK(s.minScore >= previous.minScore);                                            // :2469
			s.hmm = s.hmm.next(s.minScore - previous.minScore, W);
			s.numCells = cells;
			return s;
		}
	}
	// the band has 200 000 cells or more: cell by cell, only where the score stays within the bandwidth of the row's minimum (:2499-2520)
	Slice s(&dense);
	s.j = previous.j + W;
	s.hmm = previous.hmm;
	NodeCalc r = fillSliceSparse(seq, s.j, s.cells, previous, processed, bandwidth);
	s.cellsProcessed = r.cellsProcessed;
	s.minIndex = r.minIndex;
	s.minScore = r.minScore;
	GAO_CHECK(s.minScore >= previous.minScore);
	s.hmm = s.hmm.next(s.minScore - previous.minScore, W);
	finishSparseSlice(s, curBand, (int)seq.size(), bandwidth);
	s.sparse = true;
	if (countSparse) statSparse++;
	return s;
}

// ------------------------------------------------------------------------------------------
// the sparse method (calculateSliceAlternate, GraphAligner.h:2148-2329; setValue :2130-2146)
// ------------------------------------------------------------------------------------------
NodeCalc Engine::fillSliceSparse(const std::string& seq, size_t startj, Store& cur, const Slice& previous, std::vector<bool>& processed, int bandwidth) const
{
	typedef std::pair<size_t, size_t> Cell;                                                       // (node, column)
	// With a bandwidth of 0 (slice 0 of a run without -B: the slice-0 quirk :2612) the reference indexes calculables[1] of a
	// one-element vector (:2220, :2318): undefined behaviour there, reported as an assertion here.
	if (bandwidth < 1) fail(ASSERTION, "sparse method with bandwidth 0: undefined behaviour in the reference (GraphAligner.h:2220)");
	std::vector<std::vector<Cell>> now((size_t)bandwidth + 1), next((size_t)bandwidth + 1);
	const int prevMin = previous.minScore;
	const int uninitialized = (int)seq.size();
	auto put = [&](std::vector<std::vector<Cell>>& q, long idx, size_t node, size_t column) {
		GAO_CHECK(idx >= 0 && idx <= (long)bandwidth);                                             // (an out-of-range bucket would be undefined behaviour in the reference)
		q[(size_t)idx].emplace_back(node, column);
	};
	auto matchAt = [&](size_t row, size_t column) { return charMatch(seq[row], g.base(column)); };
	// cells of row startj reached from the previous slice's last row (:2163-2219), in the previous slice's container order
	previous.cells.forEach([&](size_t node, const Span& sp) {
		const size_t start = g.nodeBegin(node), len = sp.hi - sp.lo;
		auto usable = [&](size_t i) { const Column c = previous.cells.get(sp.lo + i); return c.end < prevMin + bandwidth && c.endExists; };
		auto endOf = [&](size_t i) { return previous.cells.get(sp.lo + i).end; };
		if (startj == 0)
		{
			for (size_t i = 0; i < len; i++)
			{
				if (!usable(i)) continue;
				put(now, endOf(i) - prevMin + (matchAt(startj, start + i) ? 0 : 1), node, start + i);
			}
		}
		else
		{
			for (size_t i = 0; i + 1 < len; i++)
			{
				if (!usable(i)) continue;
				GAO_CHECK(endOf(i) >= prevMin);
				put(now, endOf(i) - prevMin + 1, node, start + i);
				put(now, endOf(i) - prevMin + (matchAt(startj, start + i + 1) ? 0 : 1), node, start + i + 1);
			}
			if (usable(len - 1))
			{
				put(now, endOf(len - 1) - prevMin + 1, node, start + len - 1);
				for (size_t nb : g.out[node])
				{
					const size_t u = g.nodeBegin(nb);
					put(now, endOf(len - 1) - prevMin + (matchAt(startj, u) ? 0 : 1), nb, u);
				}
			}
		}
	});
	GAO_CHECK(now[0].size() > 0 || now[1].size() > 0);                                             // :2220
	std::vector<size_t> done;
	size_t cellsProcessed = 0;
	int minScore = prevMin;
	for (int j = 0; j < W; j++)
	{
		const long plus = now[0].size() == 0 ? -1 : 0;                                             // the row's minimum is one above the last row's
		for (int scoreplus = 0; scoreplus < bandwidth; scoreplus++)
		{
			// (the bucket can grow while it is walked: cells reached horizontally at the same score go to scoreplus + 1, never to this one)
			for (size_t k = 0; k < now[(size_t)scoreplus].size(); k++)
			{
				const Cell cell = now[(size_t)scoreplus][k];
				if (processed[cell.second]) continue;
				cellsProcessed++;
				processed[cell.second] = true;
				done.push_back(cell.second);
				const size_t nodeStart = g.nodeBegin(cell.first), nodeEnd = g.nodeEnd(cell.first);
				GAO_CHECK(cell.second >= nodeStart);
				GAO_CHECK(cell.second < nodeEnd);
				if (!cur.has(cell.first))                                                          // setValue (:2130-2146): first touch of the node
				{
					cur.addNode(cell.first, g.nodeLen(cell.first));
					const Span sp = cur.span(cell.first);
					for (size_t i = sp.lo; i < sp.hi; i++)
					{
						Column c;
						c.vp = 0; c.vn = 0; c.end = uninitialized; c.before = uninitialized; c.rows = 0; c.partial = false; c.beforeExists = false; c.endExists = true;
						cur.at(i) = c;
					}
				}
				Column& word = cur.at(cur.span(cell.first).lo + (cell.second - nodeStart));
				setCell(word, j, minScore + scoreplus);
				GAO_CHECK(columnValue(word, j) == minScore + scoreplus);                           // :2258
				put(next, scoreplus + 1 + plus, cell.first, cell.second);
				auto onward = [&](size_t node, size_t u) {
					if (!processed[u]) put(now, scoreplus + 1, node, u);
					if (j < W - 1) put(next, scoreplus + plus + (matchAt(startj + (size_t)j + 1, u) ? 0 : 1), node, u);
				};
				if (cell.second + 1 == nodeEnd) { for (size_t nb : g.out[cell.first]) onward(nb, g.nodeBegin(nb)); }
				else onward(cell.first, cell.second + 1);
			}
		}
		if (now[0].size() == 0) minScore++;
		for (size_t c : done) { GAO_CHECK(processed[c]); processed[c] = false; }
		done.clear();
		if (j < W - 1)
		{
			std::swap(now, next);
			for (auto& b : next) b.clear();
		}
	}
	if (now[0].size() == 0) std::swap(now[0], now[1]);
	GAO_CHECK(now[0].size() > 0);
	NodeCalc res;
	res.minScore = minScore;
	for (const Cell& c : now[0]) res.minIndex.push_back(c.second);
	res.cellsProcessed = cellsProcessed;
	return res;
}

// finalizeAlternateSlice (GraphAligner.h:2523-2552)
void Engine::finishSparseSlice(Slice& s, std::vector<bool>& curBand, int uninitialized, int bandwidth) const
{
	std::vector<std::pair<size_t, Span>> touched;
	s.cells.forEach([&](size_t node, const Span& sp) { touched.emplace_back(node, sp); });
	for (const auto& t : touched)
	{
		const size_t node = t.first;
		const Span sp = t.second;
		s.nodes.push_back(node);
		GAO_CHECK(!curBand[node]);
		curBand[node] = true;
		int minScore = s.cells.at(sp.lo).end;
		for (size_t i = sp.lo; i < sp.hi; i++)
		{
			Column& c = s.cells.at(i);
			GAO_CHECK(c.rows <= W - 1);
			GAO_CHECK(c.rows >= 0);
			c.endExists = c.rows == W - 1;
			c.rows = W;
			c.partial = false;
			minScore = std::min(minScore, c.end);
		}
		const int fill = minScore + (int)(sp.hi - sp.lo) + bandwidth + 1;
		for (size_t i = sp.lo; i < sp.hi; i++)
		{
			Column& c = s.cells.at(i);
			if (c.end == uninitialized) { c.end = fill; c.before = fill; }
		}
		s.numCells += sp.hi - sp.lo;
		s.cells.setMin(node, minScore);
	}
}

// ------------------------------------------------------------------------------------------
// first pass with sqrt checkpoints (GraphAligner.h:2571-2856)
// ------------------------------------------------------------------------------------------
Table Engine::firstPass(const std::string& seq, const Slice& initial, size_t numSlices, size_t samplingFrequency, std::vector<Span>& dense)
{
	GAO_CHECK(initial.j == (size_t)-W);
	GAO_CHECK(initial.j + numSlices * W <= seq.size());
	Table table;
	table.samplingFrequency = samplingFrequency;
	std::vector<bool> prevBand(g.nodeCount(), false), curBand(g.nodeCount(), false);
	std::vector<size_t> compOf(g.nodeCount(), std::numeric_limits<size_t>::max());
	WorkStack work(g.nodeCount());
	for (size_t n : initial.nodes) prevBand[n] = true;
	lastRowMin = 0;
	Slice last = initial.frozenEnds();
	Slice store = last;
	GAO_CHECK(last.hmm.currentlyCorrect());
	Slice rampSlice = last;
	std::vector<bool> processed(g.bp(), false);
	size_t rampRedoIndex = (size_t)-1;
	size_t rampUntil = 0;
	size_t lastProcessed = 0;
	// the window of slices of >= 200 000 cells whose traceback is worked out while they are in memory (:2604-2606)
	Slice overridePreslice = last;
	std::vector<Slice> overrideTemps;
	bool overriding = false;
	for (size_t slice = 0; slice < numSlices; slice++)
	{
		int bandwidth = (rampUntil >= slice) ? rampBandwidth : initialBandwidth;                 // :2612 (slice 0 uses the ramp width)
		lastProcessed = slice;
		lastRowMin = last.minScore;
		Slice fresh = extendAndFill(seq, last, prevBand, curBand, compOf, work, processed, dense, bandwidth);
		if (rampUntil == slice && fresh.numCells >= kCutoff) rampUntil++;                         // :2626-2629
		if ((rampUntil == slice - 1 || (rampUntil < slice && fresh.hmm.currentlyCorrect() && fresh.hmm.falseFromCorrect)) && last.numCells < kCutoff)
		{
			rampSlice = last;
			rampRedoIndex = slice - 1;
		}
		GAO_CHECK(fresh.j == last.j + W);
		statColumns += fresh.numCells;
		statSlices += 1;
		if (record)
		{
			SliceRecord rec;
			rec.direction = recordDirection; rec.j = fresh.j; rec.bandwidth = bandwidth; rec.nodes = fresh.nodes;
			for (size_t n : fresh.nodes) { Span sp = fresh.cells.span(n); for (size_t i = sp.lo; i < sp.hi; i++) rec.columns.push_back(fresh.cells.get(i)); }
			rec.minScore = fresh.minScore; rec.minIndex = fresh.minIndex; rec.sparse = fresh.sparse;
			record->push_back(std::move(rec));
		}
		if (!fresh.hmm.correctFromCorrect)
		{
			fresh.cells.releaseDense();
			lastProcessed = slice - 1;
			break;
		}
		if (!fresh.hmm.currentlyCorrect() && rampUntil < slice && rampBandwidth > initialBandwidth)
		{
			for (size_t n : fresh.nodes) { GAO_CHECK(curBand[n]); curBand[n] = false; }
			for (size_t n : last.nodes) { GAO_CHECK(prevBand[n]); prevBand[n] = false; }
			fresh.cells.releaseDense();
			rampUntil = slice;
			std::swap(slice, rampRedoIndex);
			std::swap(last, rampSlice);
			for (size_t n : last.nodes) { GAO_CHECK(!prevBand[n]); prevBand[n] = true; }
			while (table.bandwidthPerSlice.size() > slice + 1) table.bandwidthPerSlice.pop_back();
			while (table.correctness.size() > slice + 1) table.correctness.pop_back();
			while (table.slices.size() > 1 && table.slices.back().j > slice * W) table.slices.pop_back();
			if (overriding)                                                                        // :2673-2700
			{
				if (overridePreslice.j > last.j) { overriding = false; overrideTemps.clear(); }
				else
				{
					// "shorten": the reference swaps an empty slice (j = SIZE_MAX) into the back of the list without popping it and
					// tests the back's j again -- it never leaves that loop once it has entered it (:2690-2694)
					if (overrideTemps.size() > 0 && overrideTemps.back().j > last.j)
						fail(ASSERTION, "ramp redo inside a backtrace-override window: the reference does not terminate (GraphAligner.h:2690-2694)");
				}
			}
			while (table.overrides.size() > 0 && table.overrides.back().endj > last.j) table.overrides.pop_back();
			continue;
		}
		if (!overriding && fresh.numCells >= kCutoff && last.numCells < kCutoff)                  // :2721-2764
		{
			overridePreslice = last;
			overriding = true;
			overrideTemps.push_back(fresh.frozenFull());
		}
		else if (overriding)
		{
			if (fresh.numCells < kCutoff)
			{
				GAO_CHECK(overrideTemps.size() > 0);
				GAO_CHECK(last.j == overrideTemps.back().j);
				table.overrides.push_back(makeOverride(seq, overridePreslice, overrideTemps));
				statOverrides++;
				overriding = false;
				while (table.slices.size() > 0 && table.slices.back().j >= table.overrides.back().startj && table.slices.back().j <= table.overrides.back().endj) table.slices.pop_back();
				table.slices.push_back(last);
				store = fresh.frozenEnds();
				overrideTemps.clear();
			}
			else overrideTemps.push_back(fresh.frozenFull());
		}
		GAO_CHECK(table.bandwidthPerSlice.size() == slice);
		table.bandwidthPerSlice.push_back(bandwidth);
		table.correctness.push_back(fresh.hmm);
		if (slice % samplingFrequency == 0)
		{
			if (table.slices.size() == 0 || store.j != table.slices.back().j)
			{
				table.slices.push_back(store);
				store = fresh.frozenEnds();
			}
		}
		if (fresh.estimatedMemory() < store.estimatedMemory()) store = fresh.frozenEnds();       // cheapest slice of the window (:2783-2786)
		for (size_t n : last.nodes) { GAO_CHECK(prevBand[n]); prevBand[n] = false; }
		GAO_CHECK(fresh.minScore >= last.minScore);
		last = fresh.frozenEnds();
		fresh.cells.releaseDense();
		std::swap(prevBand, curBand);
	}
	if (overriding)                                                                               // :2810-2825
	{
		GAO_CHECK(overrideTemps.size() > 0);
		GAO_CHECK(last.j == overrideTemps.back().j);
		table.overrides.push_back(makeOverride(seq, overridePreslice, overrideTemps));
		statOverrides++;
		overriding = false;
		overrideTemps.clear();
		while (table.slices.size() > 0 && table.slices.back().j >= table.overrides.back().startj && table.slices.back().j <= table.overrides.back().endj) table.slices.pop_back();
	}
	GAO_CHECK(table.bandwidthPerSlice.size() == lastProcessed + 1);                             // :2833
	GAO_CHECK(table.slices.size() > 0);
	for (size_t i = 0; i < table.slices.size(); i++) GAO_CHECK(i <= 1 || table.slices[i].j > table.slices[i - 1].j);
	for (size_t i = 1; i < table.slices.size(); i++) GAO_CHECK(table.slices[i].minScore >= table.slices[i - 1].minScore);
	for (size_t i = 0; i < table.overrides.size(); i++) GAO_CHECK(table.overrides[i].endj >= table.overrides[i].startj);
	for (size_t i = 1; i < table.overrides.size(); i++) GAO_CHECK(table.overrides[i].startj > table.overrides[i - 1].endj);
	return table;
}

// ------------------------------------------------------------------------------------------
// BacktraceOverride (GraphAligner.h:167-354)
// ------------------------------------------------------------------------------------------
Override Engine::makeOverride(const std::string& seq, const Slice& previous, const std::vector<Slice>& window) const
{
	GAO_CHECK(window.size() > 0);
	Override o;
	o.startj = window[0].j;
	o.endj = window.back().j;
	GAO_CHECK(o.endj == o.startj + (window.size() - 1) * W);
	const size_t nRows = W * window.size();
	o.items.resize(nRows);
	std::vector<std::unordered_map<size_t, size_t>> indexOf(nRows);
	auto endExistsAt = [&](size_t row, size_t column) {
		const Slice& s = window[row / W];
		const size_t node = g.nodeOf(column);
		GAO_CHECK(s.cells.has(node));
		return s.cells.get(s.cells.span(node).lo + (column - g.nodeBegin(node))).endExists;
	};
	auto predecessorOf = [&](Pos pos, size_t row) {
		const size_t si = row / W;
		return si > 0 ? pickPredecessor(seq, window[si], pos, window[si - 1]) : pickPredecessor(seq, window[0], pos, previous);
	};
	// every cell reachable backwards from an existing end cell (addReachableRec :236-267; the recursion as a loop: a cell has one predecessor)
	auto reach = [&](Pos pos, size_t row) {
		while (true)
		{
			GAO_CHECK(row < nRows);
			if (indexOf[row].count(pos.first) == 1) return;
			const size_t size = indexOf[row].size();
			indexOf[row][pos.first] = size;
			if (row > 0 && row % W == W - 1 && !endExistsAt(row, pos.first)) return;
			GAO_CHECK(row == pos.second - window[0].j);
			const Pos pred = predecessorOf(pos, row);
			GAO_CHECK(pred.second == pos.second || pred.second == pos.second - 1);
			if (!(pred.second >= window[0].j && pred.second != (size_t)-1)) return;
			pos = pred;
			row = pred.second - window[0].j;
		}
	};
	{
		const Slice& bottom = window.back();
		const size_t endRow = bottom.j + W - 1;
		bottom.cells.forEach([&](size_t node, const Span& sp) {
			const size_t start = g.nodeBegin(node);
			for (size_t i = 0; i < sp.hi - sp.lo; i++) if (bottom.cells.get(sp.lo + i).endExists) reach(Pos{start + i, endRow}, nRows - 1);
		});
	}
	for (size_t row = nRows; row-- > 0;)                                                          // makeTrace :293-341
	{
		o.items[row].resize(indexOf[row].size());
		for (const auto& kv : indexOf[row])
		{
			const size_t w = kv.first, index = kv.second;
			Override::Item& item = o.items[row][index];
			const Pos pos{w, window[0].j + row};
			item.pos = pos;
			if (row % W == W - 1 && !endExistsAt(row, w)) { item.end = true; continue; }
			const Pos pred = predecessorOf(pos, row);
			if (pred.second == pos.second)
			{
				item.sameRow = true;
				auto it = indexOf[row].find(pred.first);
				if (it == indexOf[row].end()) fail(ASSERTION, "override: predecessor not indexed");    // unordered_map::at would throw
				item.previous = it->second;
			}
			else
			{
				item.sameRow = false;
				if (row != 0)
				{
					auto it = indexOf[row - 1].find(pred.first);
					if (it == indexOf[row - 1].end()) fail(ASSERTION, "override: predecessor not indexed");
					item.previous = it->second;
				}
				else item.previous = pred.first;
			}
		}
		for (const auto& item : o.items[row]) GAO_CHECK(item.end || item.pos.first != 0);
	}
	return o;
}

// BacktraceOverride::GetBacktrace (:196-231): backwards from `start` (a cell of the window's last row) to the row above the window
std::vector<Pos> Engine::overrideBacktrace(const Override& o, Pos start) const
{
	GAO_CHECK(o.items.size() > 0);
	GAO_CHECK(o.items.size() % W == 0);
	GAO_CHECK(o.items.back().size() > 0);
	GAO_CHECK(o.items.back()[0].pos.second == start.second);
	size_t index = (size_t)-1, row = o.items.size() - 1;
	for (size_t i = 0; i < o.items.back().size(); i++) if (o.items.back()[i].pos == start) { index = i; break; }
	GAO_CHECK(index != (size_t)-1);
	std::vector<Pos> out;
	while (true)
	{
		const Override::Item& cur = o.items[row][index];
		GAO_CHECK(!cur.end);
		out.push_back(cur.pos);
		const size_t nextRow = cur.sameRow ? row : row - 1;
		if (nextRow == (size_t)-1)
		{
			out.emplace_back(cur.previous, cur.pos.second - 1);
			break;
		}
		index = cur.previous;
		row = nextRow;
	}
	return out;
}

// ------------------------------------------------------------------------------------------
// recompute the slices after checkpoint `startIndex`, keeping full bits (GraphAligner.h:2858-2943)
// ------------------------------------------------------------------------------------------
std::vector<Slice> Engine::recompute(const std::string& seq, size_t overrideLastJ, const Table& table, size_t startIndex, std::vector<Span>& dense)
{
	GAO_CHECK(startIndex < table.slices.size());
	size_t startSlice = (table.slices[startIndex].j + W) / W;
	GAO_CHECK(overrideLastJ > startSlice * W);
	size_t endSlice = startIndex == table.slices.size() - 1 ? table.bandwidthPerSlice.size() : (table.slices[startIndex + 1].j + W) / W;
	if (endSlice * W >= overrideLastJ) endSlice = overrideLastJ / W;
	GAO_CHECK(endSlice > startSlice);
	GAO_CHECK(endSlice <= table.bandwidthPerSlice.size());
	const Slice& initial = table.slices[startIndex];
	std::vector<Slice> out;
	std::vector<bool> prevBand(g.nodeCount(), false), curBand(g.nodeCount(), false);
	std::vector<size_t> compOf(g.nodeCount(), std::numeric_limits<size_t>::max());
	WorkStack work(g.nodeCount());
	std::vector<bool> processed(g.bp(), false);
	for (size_t n : initial.nodes) prevBand[n] = true;
	lastRowMin = 0;
	Slice last = initial.frozenEnds();
	std::vector<SliceRecord>* keep = record;
	record = nullptr;
	countSparse = false;
	size_t keepCols = statColumns, keepSlices = statSlices;
	for (size_t slice = startSlice; slice < endSlice; slice++)
	{
		int bandwidth = (int)table.bandwidthPerSlice[slice];
		lastRowMin = last.minScore;
		Slice fresh = extendAndFill(seq, last, prevBand, curBand, compOf, work, processed, dense, bandwidth);
		GAO_CHECK(out.size() == 0 || fresh.j == out.back().j + W);
		out.push_back(fresh.frozenFull());
		for (size_t n : last.nodes) { GAO_CHECK(prevBand[n]); prevBand[n] = false; }
		GAO_CHECK(fresh.minScore >= last.minScore);
		last = fresh.frozenEnds();
		fresh.cells.releaseDense();
		std::swap(prevBand, curBand);
	}
	record = keep;
	countSparse = true;
	statColumns = keepCols; statSlices = keepSlices;
	for (size_t i = 1; i < out.size(); i++) GAO_CHECK(out[i].minScore >= out[i - 1].minScore);
	return out;
}

void Engine::trimWrongEnd(Table& t) const
{
	// GraphAligner.h:2554-2569
	bool ok = t.correctness.back().currentlyCorrect();
	while (!ok)
	{
		t.correctness.pop_back();
		t.bandwidthPerSlice.pop_back();
		if (t.correctness.size() == 0) break;
		ok = t.correctness.back().falseFromCorrect;
	}
	if (t.correctness.size() == 0) t.slices.clear();
	while (t.slices.size() > 1 && t.slices.back().j >= t.correctness.size() * W) t.slices.pop_back();
}

Slice Engine::seedSlice(size_t node) const
{
	// GraphAligner.h:2945-2960
	Slice s;
	s.j = (size_t)-W;
	s.cells.addNode(node, g.nodeLen(node));
	s.cells.setMin(node, 0);
	s.minScore = 0;
	s.minIndex.push_back(g.nodeEnd(node) - 1);
	s.nodes.push_back(node);
	Span sp = s.cells.span(node);
	for (size_t i = sp.lo; i < sp.hi; i++)
	{
		Column c;
		c.rows = W;
		s.cells.at(i) = c;
	}
	return s;
}

Engine::Split Engine::splitAlign(const std::string& sequence, int bigraphNode, bool backwards, size_t pos, std::vector<Span>& dense)
{
	// GraphAligner.h:2969-3024
	GAO_CHECK(pos < sequence.size());
	auto find = [&](int id) { auto it = g.lookup.find(id); if (it == g.lookup.end()) fail(BAD_SEED, "seed node not in graph"); return it->second; };
	size_t fwNode = find(backwards ? bigraphNode * 2 + 1 : bigraphNode * 2);
	size_t bwNode = find(backwards ? bigraphNode * 2 : bigraphNode * 2 + 1);
	GAO_CHECK(g.nodeLen(fwNode) == g.nodeLen(bwNode));
	Split res;
	res.splitIndex = pos;
	auto padded = [](std::string s) { size_t pad = (W - (s.size() % W)) % W; s.append(pad, 'N'); return s; };
	auto frequency = [](size_t len) { return (size_t)(int)sqrt(len / W); };                     // :2962-2967
	if (pos > 0)
	{
		GAO_CHECK(sequence.size() >= pos + g.dbgOverlap);
		std::string part = padded(reverseComplement(sequence.substr(0, pos + g.dbgOverlap)));
		recordDirection = 1;
		Table t = firstPass(part, seedSlice(bwNode), part.size() / W, frequency(part.size()), dense);
		trimWrongEnd(t);
		res.backward = std::move(t);
	}
	if (pos < sequence.size() - 1)
	{
		std::string part = padded(sequence.substr(pos));
		recordDirection = 0;
		Table t = firstPass(part, seedSlice(fwNode), part.size() / W, frequency(part.size()), dense);
		trimWrongEnd(t);
		res.forward = std::move(t);
	}
	return res;
}

// ------------------------------------------------------------------------------------------
// traceback (GraphAligner.h:894-1021)
// ------------------------------------------------------------------------------------------
std::vector<Pos> Engine::traceInSlice(const std::string& seq, const Slice& s, Pos pos) const
{
	std::vector<Pos> out;
	while (pos.second != s.j)
	{
		pos = pickPredecessor(seq, s, pos, s);
		out.push_back(pos);
	}
	return out;
}

std::vector<Pos> Engine::traceBoundary(const std::string& seq, const Slice& after, const Slice& before, size_t column) const
{
	Pos pos{column, after.j};
	GAO_CHECK(after.j == before.j + W);
	std::vector<Pos> out;
	while (pos.second == after.j)
	{
		pos = pickPredecessor(seq, after, pos, before);
		out.push_back(pos);
	}
	GAO_CHECK(before.cells.has(g.nodeOf(pos.first)));
	return out;
}

std::vector<Pos> Engine::traceInner(const std::string& seq, const std::vector<Slice>& part, Pos pos) const
{
	GAO_CHECK(part.size() > 0);
	std::vector<Pos> out;
	out.push_back(pos);
	for (size_t s = part.size(); s-- > 0;)
	{
		GAO_CHECK(part[s].j <= out.back().second);
		GAO_CHECK(part[s].j + W > out.back().second);
		auto inSlice = traceInSlice(seq, part[s], out.back());
		GAO_CHECK(inSlice.size() >= (size_t)W - 1);
		out.insert(out.end(), inSlice.begin(), inSlice.end());
		if (s > 0)
		{
			auto across = traceBoundary(seq, part[s], part[s - 1], out.back().first);
			out.insert(out.end(), across.begin(), across.end());
		}
	}
	GAO_CHECK(out.back().second == part[0].j);
	return out;
}

Engine::Trace Engine::traceTable(const std::string& seq, const Table& table, std::vector<Span>& dense)
{
	const int big = std::numeric_limits<int>::max();
	GAO_CHECK(table.bandwidthPerSlice.size() == table.correctness.size());
	GAO_CHECK(seq.size() % W == 0);
	if (table.slices.size() == 0) return Trace{big, {}};
	if (table.bandwidthPerSlice.size() == 0) return Trace{big, {}};
	GAO_CHECK(table.samplingFrequency > 1);                                                    // :906 (reads shorter than 193 bp per direction fail here)
	Trace result{0, {}};
	size_t overrideIndex = (size_t)-1, lastOverrideStartJ = (size_t)-1, nextOverrideEndJ = (size_t)-1;
	if (table.overrides.size() > 0)
	{
		overrideIndex = table.overrides.size() - 1;
		nextOverrideEndJ = table.overrides.back().endj;
	}
	for (size_t i = table.slices.size(); i-- > 0;)
	{
		if ((table.slices[i].j + W) / W == table.bandwidthPerSlice.size())
		{
			GAO_CHECK(i == table.slices.size() - 1);
			result.first = table.slices.back().minScore;
			result.second.emplace_back(table.slices.back().minIndex.back(), table.slices.back().j + W - 1);
			continue;
		}
		auto part = recompute(seq, lastOverrideStartJ, table, i, dense);
		GAO_CHECK(part.size() > 0);
		if (i == table.slices.size() - 1)
		{
			result.first = part.back().minScore;
			GAO_CHECK(part.back().minIndex.size() > 0);
			result.second.emplace_back(part.back().minIndex.back(), part.back().j + W - 1);
		}
		auto inner = traceInner(seq, part, result.second.back());
		GAO_CHECK(inner.size() > 1);
		result.second.insert(result.second.end(), inner.begin() + 1, inner.end());
		auto across = traceBoundary(seq, part[0], table.slices[i], result.second.back().first);
		result.second.insert(result.second.end(), across.begin(), across.end());
		GAO_CHECK(across.size() > 0);
		if (table.slices[i].j == nextOverrideEndJ)
		{
			// checkpoint i is the last slice of a window whose traceback was worked out in the first pass (:939-946)
			auto through = overrideBacktrace(table.overrides[overrideIndex], result.second.back());
			statOverrideTraces++;
			result.second.insert(result.second.end(), through.begin() + 1, through.end());
			lastOverrideStartJ = table.overrides[overrideIndex].startj;
			overrideIndex--;
			if (overrideIndex != (size_t)-1) nextOverrideEndJ = table.overrides[overrideIndex].endj;
		}
	}
	GAO_CHECK(result.second.back().second == (size_t)-1);
	result.second.pop_back();
	GAO_CHECK(result.second.back().second == 0);
	std::reverse(result.second.begin(), result.second.end());
	GAO_CHECK(result.second[0].second == 0);                                                    // verifyTrace (:853)
	return result;
}

std::pair<Engine::Trace, Engine::Trace> Engine::piecewise(const Split& split, const std::string& sequence, std::vector<Span>& dense)
{
	// GraphAligner.h:3039-3098
	GAO_CHECK(split.splitIndex < sequence.size());
	Trace fw{0, {}}, bw{0, {}};
	auto padded = [](std::string s) { size_t pad = (W - (s.size() % W)) % W; s.append(pad, 'N'); return s; };
	if (split.splitIndex < sequence.size() - 1 && split.forward.slices.size() > 0)
	{
		GAO_CHECK(sequence.size() >= split.splitIndex + g.dbgOverlap);
		size_t traceable = sequence.size() - split.splitIndex - g.dbgOverlap;
		std::string part = padded(sequence.substr(split.splitIndex));
		fw = traceTable(part, split.forward, dense);
		while (fw.second.size() > 0 && fw.second.back().second >= traceable) fw.second.pop_back();
	}
	if (split.splitIndex > 0 && split.backward.slices.size() > 0)
	{
		GAO_CHECK(sequence.size() >= split.splitIndex + g.dbgOverlap);
		size_t traceable = split.splitIndex;
		std::string part = padded(reverseComplement(sequence.substr(0, split.splitIndex + g.dbgOverlap)));
		bw = traceTable(part, split.backward, dense);
		while (bw.second.size() > 0 && bw.second.back().second >= traceable) bw.second.pop_back();
		// reverseTrace (:3026-3037)
		if (bw.second.size() > 0)
		{
			std::reverse(bw.second.begin(), bw.second.end());
			size_t endRow = split.splitIndex - 1;
			for (auto& p : bw.second)
			{
				p.first = g.reverseColumn(p.first);
				GAO_CHECK(p.second <= endRow);
				p.second = endRow - p.second;
			}
		}
		// the forward rows are shifted only inside this branch (:3091-3094)
		for (auto& p : fw.second) p.second += split.splitIndex;
	}
	return std::make_pair(fw, bw);
}

void Engine::noteTried(std::vector<std::tuple<size_t, size_t, size_t>>& tried, const std::pair<Trace, Trace>& tr) const
{
	// addAlignmentNodes (GraphAligner.h:594-634)
	for (const std::vector<Pos>* t : {&tr.first.second, &tr.second.second})
	{
		if (t->size() == 0) continue;
		size_t oldNode = g.nodeOf((*t)[0].first);
		size_t lo = (*t)[0].second, hi = (*t)[0].second;
		for (size_t i = 1; i < t->size(); i++)
		{
			size_t n = g.nodeOf((*t)[i].first);
			size_t row = (*t)[i].second;
			if (n != oldNode)
			{
				tried.emplace_back(lo, hi, oldNode);
				lo = row;
				oldNode = n;
			}
			hi = row;
		}
		tried.emplace_back(lo, hi, oldNode);
	}
}

std::vector<TraceItem> Engine::traceItemsInner(const std::string& seq, const std::vector<Pos>& tr) const
{
	// getTraceInfoInner (GraphAligner.h:718-780)
	std::vector<TraceItem> out;
	for (size_t i = 1; i < tr.size(); i++)
	{
		Pos now = tr[i], old = tr[i - 1];
		GAO_CHECK(now.second == old.second || now.second == old.second + 1);
		GAO_CHECK(now.second != old.second || now.first != old.first);
		size_t oldNode = g.nodeOf(old.first), newNode = g.nodeOf(now.first);
		if (old.first == g.nodeEnd(oldNode) - 1) GAO_CHECK(now.first == old.first || now.first == g.nodeBegin(newNode));
		else GAO_CHECK(now.first == old.first || now.first == old.first + 1);
		bool diagonal = now.second != old.second;
		if (now.first == old.first)
		{
			bool selfLoop = now.second == old.second + 1 && g.nodeLen(newNode) == 1 &&
				std::find(g.out[newNode].begin(), g.out[newNode].end(), newNode) != g.out[newNode].end();
			if (!selfLoop) diagonal = false;
		}
		TraceItem it;
		it.nodeID = g.ids[newNode] / 2;
		it.reverse = g.ids[newNode] % 2 == 1;
		it.offset = now.first - g.nodeBegin(newNode);
		it.readpos = now.second;
		it.graphChar = g.base(now.first);
		it.readChar = seq[now.second];
		if (now.second == old.second) it.type = DELETION;
		else if (now.first == old.first && !diagonal) it.type = INSERTION;
		else it.type = charMatch(seq[now.second], g.base(now.first)) ? MATCH : MISMATCH;
		out.push_back(it);
	}
	return out;
}

std::vector<TraceItem> Engine::traceItems(const std::string& seq, const std::vector<Pos>& bw, const std::vector<Pos>& fw) const
{
	// getTraceInfo (GraphAligner.h:690-716)
	std::vector<TraceItem> out;
	if (bw.size() > 0) { auto v = traceItemsInner(seq, bw); out.insert(out.end(), v.begin(), v.end()); }
	if (bw.size() > 0 && fw.size() > 0)
	{
		size_t n = g.nodeOf(fw[0].first);
		TraceItem it;
		it.type = FORWARDBACKWARDSPLIT;
		it.nodeID = g.ids[n] / 2;
		it.reverse = n % 2 == 1;                                                                 // node INDEX parity, as in the reference (:704)
		it.offset = fw[0].first - g.nodeBegin(n);
		it.readpos = fw[0].second;
		it.graphChar = g.base(fw[0].first);
		it.readChar = seq[fw[0].second];
		out.push_back(it);
	}
	if (fw.size() > 0) { auto v = traceItemsInner(seq, fw); out.insert(out.end(), v.begin(), v.end()); }
	return out;
}

Engine::Partial Engine::toMappings(const std::string& sequence, int score, const std::vector<Pos>& trace) const
{
	// traceToAlignment (GraphAligner.h:782-847)
	Partial res;
	res.score = score;
	res.failed = false;
	if (trace.size() == 0) { res.failed = true; return res; }
	size_t pos = 0;
	size_t oldNode = g.nodeOf(trace[0].first);
	while (oldNode == g.dummyFirst)
	{
		pos++;
		if (pos == trace.size()) { res.failed = true; res.score = std::numeric_limits<int32_t>::max(); return res; }
		GAO_CHECK(trace[pos].second >= trace[pos - 1].second);
		oldNode = g.nodeOf(trace[pos].first);
	}
	if (oldNode == g.dummyLast) { res.failed = true; res.score = std::numeric_limits<int32_t>::max(); return res; }
	int rank = 0;
	Mapping m;
	m.rank = rank; m.nodeId = g.ids[oldNode]; m.isReverse = g.rev[oldNode]; m.offset = (int64_t)(trace[pos].first - g.nodeBegin(oldNode));
	Pos nodeStart = trace[pos], nodeEnd = trace[pos], beforeNode = trace[pos];
	for (; pos < trace.size(); pos++)
	{
		if (g.nodeOf(trace[pos].first) == g.dummyLast) break;   // (sic) a column index compared with a node index, as in the reference (:802,816)
		if (g.nodeOf(trace[pos].first) == oldNode) { nodeEnd = trace[pos]; continue; }
		GAO_CHECK(trace[pos].second >= trace[pos - 1].second);
		GAO_CHECK(nodeEnd.second >= nodeStart.second);
		GAO_CHECK(nodeEnd.first >= nodeStart.first);
		m.fromLength = (int64_t)(nodeEnd.first - nodeStart.first + 1);
		m.toLength = (int64_t)(nodeEnd.second - beforeNode.second);
		m.editSeq = sequence.substr(nodeStart.second, nodeEnd.second - beforeNode.second);
		res.mappings.push_back(m);
		oldNode = g.nodeOf(trace[pos].first);
		beforeNode = nodeEnd;
		nodeStart = trace[pos];
		nodeEnd = trace[pos];
		rank++;
		m = Mapping{};
		m.rank = rank; m.nodeId = g.ids[oldNode]; m.isReverse = g.rev[oldNode];
	}
	m.fromLength = (int64_t)(nodeEnd.first - nodeStart.first);                                  // no +1 on the last mapping (:843)
	m.toLength = (int64_t)(nodeEnd.second - beforeNode.second);
	m.editSeq = sequence.substr(nodeStart.second, nodeEnd.second - beforeNode.second);
	res.mappings.push_back(m);
	return res;
}

Engine::Partial Engine::mergePartials(const Partial& first, const Partial& second) const
{
	// mergeAlignments (GraphAligner.h:648-688)
	GAO_CHECK(!first.failed || !second.failed);
	if (first.failed) return second;
	if (second.failed) return first;
	if (first.mappings.size() == 0) return second;
	if (second.mappings.size() == 0) return first;
	Partial out;
	out.failed = false;
	out.mappings = first.mappings;
	out.score = first.score + second.score;
	size_t startAt = 0;
	const Mapping& a = first.mappings.back();
	const Mapping& b = second.mappings.front();
	size_t an = g.lookup.at((int)a.nodeId), bn = g.lookup.at((int)b.nodeId);
	if (a.nodeId == b.nodeId && a.isReverse == b.isReverse) startAt = 1;
	else if (std::find(g.out[an].begin(), g.out[an].end(), bn) != g.out[an].end()) startAt = 0;
	// else: "Piecewise alignments can't be merged!" is only logged (:676-681)
	for (size_t i = startAt; i < second.mappings.size(); i++) out.mappings.push_back(second.mappings[i]);
	return out;
}

AlignResult Engine::align(const std::string& seqId, const std::string& sequence, const std::vector<Seed>& seeds)
{
	// GraphAligner.h:408-491
	(void)seqId;
	AlignResult res;
	GAO_CHECK(g.finalized);
	GAO_CHECK(seeds.size() > 0);
	size_t bestEstimate = 0;
	Seed bestSeed;
	std::vector<std::tuple<size_t, size_t, size_t>> tried;
	std::pair<Trace, Trace> bestTrace;
	bool have = false;
	std::vector<Span> dense(g.nodeCount());
	for (size_t i = 0; i < seeds.size(); i++)
	{
		auto it = g.lookup.find(std::get<0>(seeds[i]) * 2);
		if (it == g.lookup.end()) fail(BAD_SEED, "seed node not in graph");
		size_t nodeIndex = it->second;
		size_t pos = std::get<1>(seeds[i]);
		bool covered = false;
		for (auto& t : tried) if (std::get<0>(t) <= pos && std::get<1>(t) >= pos && std::get<2>(t) == nodeIndex) { covered = true; break; }
		if (covered) continue;
		Split split = splitAlign(sequence, std::get<0>(seeds[i]), std::get<2>(seeds[i]), pos, dense);
		auto trace = piecewise(split, sequence, dense);
		noteTried(tried, trace);
		if (!have || split.estimated() > bestEstimate)
		{
			bestTrace = std::move(trace);
			bestSeed = seeds[i];
			bestEstimate = split.estimated();
			have = true;
		}
	}
	res.columnsFirstPass = statColumns;
	res.slicesFirstPass = statSlices;
	res.sparseSlices = statSparse; res.overrideWindows = statOverrides; res.overrideTraces = statOverrideTraces;
	const int big = std::numeric_limits<int>::max();
	if (!have) return res;
	if (bestTrace.first.first == big && bestTrace.second.first == big) return res;
	auto items = traceItems(sequence, bestTrace.second.second, bestTrace.first.second);
	Partial fw = toMappings(sequence, bestTrace.first.first, bestTrace.first.second);
	Partial bw = toMappings(sequence, bestTrace.second.first, bestTrace.second.second);
	if (fw.failed && bw.failed) return res;
	Partial merged = mergePartials(bw, fw);
	res.failed = false;
	res.score = merged.score;
	res.mappings = merged.mappings;
	res.trace = items;
	size_t lastAligned;
	if (bestTrace.second.second.size() > 0) lastAligned = bestTrace.second.second[0].second;
	else
	{
		lastAligned = std::get<1>(bestSeed);
		GAO_CHECK(bestTrace.first.second.size() > 0);
	}
	res.queryPosition = lastAligned;
	res.alignmentStart = lastAligned;
	res.alignmentEnd = lastAligned + bestEstimate;
	res.fwTrace = bestTrace.first.second;
	res.bwTrace = bestTrace.second.second;
	res.fwScore = bestTrace.first.first;
	res.bwScore = bestTrace.second.first;
	return res;
}

}  // namespace

std::vector<size_t> frozenIterationOrder(const std::vector<size_t>& nodes, size_t graphNodes)
{
	// what projectForwardFromMinScore iterates over: a live slice's nodes re-inserted, in band
	// order, into the frozen copy's std::unordered_map (NodeSlice.h:724-740)
	std::vector<Span> dense(graphNodes);
	Store live(&dense);
	for (size_t n : nodes) live.addNode(n, 1);
	Store frozen = live.frozenEnds();
	std::vector<size_t> out;
	frozen.forEach([&](size_t n, const Span&) { out.push_back(n); });
	return out;
}

// component hooks for the tests that pin the oracle's helpers against the reference's own (tests/test_oracle_refparts.py)
int interleavedRankForTest(uint64_t vp, uint64_t vn, int lo, int hi, int rank) { return interleavedRank(vp, vn, lo, hi, rank); }
std::vector<size_t> workStackForTest(const std::vector<long long>& ops, size_t universe)
{
	WorkStack q(universe);
	std::vector<size_t> popped;
	for (long long op : ops)
	{
		if (op >= 0) q.push((size_t)op);
		else if (q.size() > 0) { popped.push_back(q.top()); q.pop(); }
	}
	while (q.size() > 0) { popped.push_back(q.top()); q.pop(); }
	return popped;
}

AlignResult alignOneWay(const Graph& g, const std::string& seqId, const std::string& sequence, int initialBandwidth, int rampBandwidth,
                        const std::vector<Seed>& seeds, std::vector<SliceRecord>* record)
{
	try
	{
		Engine e(g, initialBandwidth, rampBandwidth, record);
		return e.align(seqId, sequence, seeds);
	}
	catch (const Failure& f)
	{
		AlignResult r;
		r.status = f.status;
		r.message = f.what;
		r.failed = true;
		return r;
	}
}

}  // namespace gao
