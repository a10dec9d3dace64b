// ga_scalegen.cpp -- TEST-ONLY helper: builds a very large synthetic graph through the product's public C ABI
// (ga_graph_add_node / ga_graph_add_edge, the calls the reference's loaders make) from native code, because a graph
// past 2^27 directed nodes cannot be fed node by node from Python in reasonable time.  The genome is a counter-based
// random sequence, so any window of it can be regenerated later (for reads and for the small oracle graph).
// Linked against graphaligner_amd/libgraphaligner_amd.so; never part of the product.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/graphaligner_amd.h"

static inline uint64_t mix(uint64_t x)
{
	x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}
// base at genome position p (32 bases per 64-bit hash word)
static inline char baseAt(uint64_t seed, uint64_t p) { return "ACGT"[(mix(seed ^ (p >> 5) * 0x2545F4914F6CDD1Dull) >> ((p & 31) * 2)) & 3]; }

extern "C" {

// bases [start, start + len) of the genome
void ga_scalegen_region(uint64_t seed, uint64_t start, uint64_t len, char* out)
{
	for (uint64_t i = 0; i < len; i++) out[i] = baseAt(seed, start + i);
}

// one chain of `n_nodes` bidirected nodes of `node_len` bp (ids 1 .. n_nodes), both strands, Finalize included.
// Node i holds genome[(i - 1) * node_len, i * node_len).  Returns a ga_status.
int ga_scalegen_chain(ga_graph_t* g, uint64_t seed, uint64_t n_nodes, int node_len)
{
	std::string fw((size_t)node_len, 'A'), rc((size_t)node_len, 'A');
	for (uint64_t i = 1; i <= n_nodes; i++)
	{
		const uint64_t at = (i - 1) * (uint64_t)node_len;
		for (int k = 0; k < node_len; k++)
		{
			const char c = baseAt(seed, at + (uint64_t)k);
			fw[(size_t)k] = c;
			rc[(size_t)(node_len - 1 - k)] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
		}
		int s = ga_graph_add_node(g, (int64_t)(2 * i), fw.data(), fw.size(), 0);          // BigraphToDigraph.cpp:29,115-116
		if (s) return s;
		s = ga_graph_add_node(g, (int64_t)(2 * i + 1), rc.data(), rc.size(), 1);
		if (s) return s;
	}
	for (uint64_t i = 1; i < n_nodes; i++)
	{
		// bidirected edge i+ -> (i+1)+ = directed 2i -> 2(i+1) and (2(i+1)+1) -> (2i+1)   (BigraphToDigraph.cpp:32-56)
		int s = ga_graph_add_edge(g, (int64_t)(2 * i), (int64_t)(2 * (i + 1)));
		if (s) return s;
		s = ga_graph_add_edge(g, (int64_t)(2 * (i + 1) + 1), (int64_t)(2 * i + 1));
		if (s) return s;
	}
	return ga_graph_finalize(g, 0);
}

}  // extern "C"

// ---- a variation graph at whole-genome scale (1000GP-like bubbles, nodes <= 32 bp) ------------------------------------
// The genome is cut into blocks of `block` bases.  A block is a chain of pieces of <= node_len bases followed by ONE variant site at its
// last base: a SNP (two 1-bp allele nodes), or -- every 11th block -- an insertion of 1-5 bases after the last piece, or -- every 13th --
// a deletion that can skip the block's last 1-5 bases.  Everything is a pure function of (seed, block index): node ids are
// block * 16 + k (pieces k = 1 .. 12, site nodes 13 and 14), so any window of the graph can be written out again as GFA for the
// oracle, and a haplotype walk can be regenerated from (seed, walk seed).
namespace {
struct Block
{
	int nPieces;                 // chain pieces 1 .. nPieces
	int pieceLen[12];
	int kind;                    // 0 SNP, 1 insertion, 2 deletion
	int siteLen;                 // SNP: 1; insertion: bases inserted; deletion: bases that can be skipped
	uint64_t start;              // first genome position of the block
};
inline Block blockAt(uint64_t seed, uint64_t b, int block, int nodeLen)
{
	Block k;
	k.start = b * (uint64_t)block;
	const uint64_t h = mix(seed * 0x9E3779B97F4A7C15ull + b * 0xD1B54A32D192ED03ull + 77);
	k.kind = b % 11 == 5 ? 1 : b % 13 == 7 ? 2 : 0;
	k.siteLen = k.kind == 0 ? 1 : 1 + (int)(h % 5);
	// chain = the block's bases minus the site's reference bases (SNP: the last base; deletion: the last siteLen bases; insertion: none)
	int chain = block - (k.kind == 0 ? 1 : k.kind == 2 ? k.siteLen : 0);
	k.nPieces = 0;
	while (chain > 0) { const int l = chain < nodeLen ? chain : nodeLen; k.pieceLen[k.nPieces++] = l; chain -= l; }
	return k;
}
inline char altBase(uint64_t seed, uint64_t pos, char ref)
{
	const char* acgt = "ACGT";
	const int r = ref == 'A' ? 0 : ref == 'C' ? 1 : ref == 'G' ? 2 : 3;
	return acgt[(r + 1 + (int)(mix(seed ^ (pos * 0x9FB21C651E98DF25ull + 3)) % 3)) % 4];
}
inline char insBase(uint64_t seed, uint64_t b, int i) { return "ACGT"[mix(seed ^ (b * 0xA24BAED4963EE407ull + 1000 + (uint64_t)i)) & 3]; }

// the nodes and edges of one block through a callback: node(id, sequence), edge(from, to); tailsIn/out = ids whose right end leads on
template <typename NodeFn, typename EdgeFn>
void emitBlock(uint64_t seed, uint64_t b, int block, int nodeLen, const int64_t* tailsIn, int nTailsIn, int64_t* tailsOut, int& nTailsOut, NodeFn node, EdgeFn edge)
{
	const Block k = blockAt(seed, b, block, nodeLen);
	const int64_t base = (int64_t)b * 16;
	std::string s;
	uint64_t at = k.start;
	int64_t last = -1;
	for (int p = 0; p < k.nPieces; p++)
	{
		s.resize((size_t)k.pieceLen[p]);
		for (int i = 0; i < k.pieceLen[p]; i++) s[(size_t)i] = baseAt(seed, at + (uint64_t)i);
		at += (uint64_t)k.pieceLen[p];
		const int64_t id = base + 1 + p;
		node(id, s);
		if (p == 0) { for (int t = 0; t < nTailsIn; t++) edge(tailsIn[t], id); } else edge(last, id);
		last = id;
	}
	if (k.kind == 0)
	{
		const char ref = baseAt(seed, at);
		node(base + 13, std::string(1, ref));
		node(base + 14, std::string(1, altBase(seed, at, ref)));
		edge(last, base + 13); edge(last, base + 14);
		tailsOut[0] = base + 13; tailsOut[1] = base + 14; nTailsOut = 2;
	}
	else if (k.kind == 1)
	{
		s.resize((size_t)k.siteLen);
		for (int i = 0; i < k.siteLen; i++) s[(size_t)i] = insBase(seed, b, i);
		node(base + 13, s);
		edge(last, base + 13);
		tailsOut[0] = last; tailsOut[1] = base + 13; nTailsOut = 2;
	}
	else
	{
		s.resize((size_t)k.siteLen);
		for (int i = 0; i < k.siteLen; i++) s[(size_t)i] = baseAt(seed, at + (uint64_t)i);
		node(base + 13, s);
		edge(last, base + 13);
		tailsOut[0] = last; tailsOut[1] = base + 13; nTailsOut = 2;
	}
}
}  // namespace

extern "C" {

// blocks [0, n_blocks) into a graph through the bigraph calls the loaders make; Finalize included
int ga_scalegen_bubbles(ga_graph_t* g, uint64_t seed, uint64_t n_blocks, int block, int node_len)
{
	if (block < 8 || block > 12 * node_len || node_len < 1) return GA_E_INVALID;
	int status = 0;
	std::vector<std::pair<int64_t, int64_t>> edges;
	edges.reserve((size_t)n_blocks * 6);
	int64_t tails[2] = {0, 0}, out[2];
	int nTails = 0, nOut = 0;
	for (uint64_t b = 0; b < n_blocks && !status; b++)
	{
		emitBlock(seed, b, block, node_len, tails, nTails, out, nOut,
		          [&](int64_t id, const std::string& s) { if (!status) status = ga_graph_add_bigraph_node(g, id, s.data(), s.size()); },
		          [&](int64_t from, int64_t to) { edges.emplace_back(from, to); });
		tails[0] = out[0]; tails[1] = out[1]; nTails = nOut;
	}
	for (const auto& e : edges) { if (status) break; status = ga_graph_add_bigraph_edge(g, e.first, 0, e.second, 0); }
	if (status) return status;
	return ga_graph_finalize(g, 0);
}

// the same blocks [b0, b1) as GFA text (S and L lines), for the oracle's small graph; returns the text's length (the caller's buffer
// must hold it: call with cap = 0 to get the length)
uint64_t ga_scalegen_bubbles_gfa(uint64_t seed, uint64_t b0, uint64_t b1, int block, int node_len, char* out, uint64_t cap)
{
	std::string text = "H\tVN:Z:1.0\n", links;
	int64_t tails[2] = {0, 0}, tout[2];
	int nTails = 0, nOut = 0;
	for (uint64_t b = b0; b < b1; b++)
	{
		emitBlock(seed, b, block, node_len, tails, nTails, tout, nOut,
		          [&](int64_t id, const std::string& s) { text += "S\t" + std::to_string(id) + "\t" + s + "\n"; },
		          [&](int64_t from, int64_t to) { links += "L\t" + std::to_string(from) + "\t+\t" + std::to_string(to) + "\t+\t0M\n"; });
		tails[0] = tout[0]; tails[1] = tout[1]; nTails = nOut;
	}
	text += links;
	if (out && cap >= text.size()) memcpy(out, text.data(), text.size());
	return text.size();
}

// a haplotype through blocks b0, b0 + 1, ... until `want` bases are out: at every site one allele by the walk's own random stream.
// Returns the number of bases written (<= want).
uint64_t ga_scalegen_bubbles_walk(uint64_t seed, uint64_t walk_seed, uint64_t b0, uint64_t n_blocks, int block, int node_len, uint64_t want, char* out)
{
	uint64_t n = 0;
	for (uint64_t b = b0; b < n_blocks && n < want; b++)
	{
		const Block k = blockAt(seed, b, block, node_len);
		uint64_t at = k.start;
		int chain = 0;
		for (int p = 0; p < k.nPieces; p++) chain += k.pieceLen[p];
		for (int i = 0; i < chain && n < want; i++) out[n++] = baseAt(seed, at + (uint64_t)i);
		at += (uint64_t)chain;
		const bool alt = (mix(walk_seed * 0x2545F4914F6CDD1Dull + b) >> 17) & 1;
		if (k.kind == 0) { if (n < want) { const char ref = baseAt(seed, at); out[n++] = alt ? altBase(seed, at, ref) : ref; } }
		else if (k.kind == 1) { if (alt) for (int i = 0; i < k.siteLen && n < want; i++) out[n++] = insBase(seed, b, i); }
		else { if (!alt) for (int i = 0; i < k.siteLen && n < want; i++) out[n++] = baseAt(seed, at + (uint64_t)i); }
	}
	return n;
}

}  // extern "C"
