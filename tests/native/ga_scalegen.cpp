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
