// ga_backend.h -- the seam between the host library (graph model, job building, result
// assembly: ga_host.cpp) and whatever executes the extension program.  The product links
// ga_device.hip (gfx950, HIP).  tests/emul/ links a host emulation of the same program so the
// device logic can be checked in the CPU-only container; that object is never part of the
// shipped library.
#pragma once
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <memory>
#include <vector>

#include "ga_types.h"

struct GaFlatGraph                      // host copy of what goes to HBM
{
	std::vector<uint64_t> node_start;   // n_nodes + 1
	std::vector<uint32_t> seq2;
	std::vector<uint32_t> in_off, in_nbr, out_off, out_nbr;
};

// the per-node records of GaDevGraph::node_rec
inline std::vector<uint32_t> ga_build_node_records(const GaFlatGraph& f)
{
	const size_t n = f.node_start.size() - 1;
	std::vector<uint32_t> rec(n * GA_NODE_REC_WORDS, 0);
	for (size_t i = 0; i < n; i++)
	{
		uint32_t* r = rec.data() + i * GA_NODE_REC_WORDS;
		r[0] = (uint32_t)f.node_start[i];
		r[1] = (uint32_t)(f.node_start[i] >> 32);
		r[2] = (uint32_t)(f.node_start[i + 1] - f.node_start[i]);
		const uint32_t inDeg = f.in_off[i + 1] - f.in_off[i], outDeg = f.out_off[i + 1] - f.out_off[i];
		r[3] = (inDeg < 0xffffu ? inDeg : 0xffffu) | ((outDeg < 0xffffu ? outDeg : 0xffffu) << 16);
		for (uint32_t k = 0; k < 4 && k < outDeg; k++) r[4 + k] = f.out_nbr[f.out_off[i] + k];
		for (uint32_t k = 0; k < 4 && k < inDeg; k++)
		{
			const uint32_t m = f.in_nbr[f.in_off[i] + k];
			r[8 + k] = m;
			r[12 + k] = (uint32_t)(f.node_start[m + 1] - f.node_start[m]);
		}
	}
	return rec;
}

struct GaRunConfig
{
	int initial_bw = 0, ramp_bw = 0;
	uint32_t max_slices = 0;            // max over jobs of n_rows / 64
	uint32_t max_rows = 0;
	uint32_t emit_runs = 0;             // 1: no result of the batch needs a cell list: the lanes = reads kernel hands back node runs instead of moves
};

struct GaRunStats
{
	double kernel_ms = 0;               // all passes of the last run
	double main_ms = 0;                 // the first pass (lanes = reads kernel)
	int main_variant = 0;               // its template arguments: band nodes per lane * 1000 + record block * 10 + (1 when 32 lanes per wave)
	uint32_t slots = 0, waves_per_cu = 0;
	uint64_t scratch_bytes = 0;
	uint64_t jobs_retried = 0;
	uint64_t stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // diagnostic builds: summed shader cycles per phase
};

// What the match words are built from when the back end builds them itself (GaBackendGraph::buildsMatchWords): the batch's copy of
// the reads and, per job, where its rows come from -- rows k < n are read characters (forward: seq[seq_off + pos + k] through rowCode;
// backward: seq[seq_off + n - 1 - k] through rowCodeRc), rows n .. padded - 1 are padCode; the job's first slice is eq_slice.
struct GaEqFill { uint64_t seq_off, eq_slice; uint32_t pos, n, padded, backward; };
struct GaEqSource
{
	const char* seq = nullptr; size_t seqBytes = 0;
	const GaEqFill* fills = nullptr; size_t nFills = 0;
	const uint8_t* rowCode = nullptr; const uint8_t* rowCodeRc = nullptr;       // 256 entries each
	uint8_t padCode = 0;
};

class GaBackendGraph
{
public:
	virtual ~GaBackendGraph() {}
	// true: ga_backend_create_batch wants a GaEqSource and builds the match words on its side (the product: a kernel over the uploaded
	// reads); false: it wants the finished words (the host emulation of tests/emul)
	virtual bool buildsMatchWords() const { return false; }
	// host memory for a batch's copy of the reads (the product: pinned blocks from a pool, so that their upload is one DMA transfer)
	virtual std::shared_ptr<char> hostBlock(size_t bytes)
	{
		void* mem = nullptr;
		const size_t two = (size_t)2 << 20;
		if (posix_memalign(&mem, two, (bytes + two - 1) & ~(two - 1)) != 0) return std::shared_ptr<char>();
		return std::shared_ptr<char>((char*)mem, [](char* p) { free(p); });
	}
};
class GaBackendBatch
{
public:
	virtual ~GaBackendBatch() {}
	// back ends that build the match words: per fill of the GaEqSource, 1 when one of its rows is a character outside IUPAC
	virtual const std::vector<uint8_t>* invalidFills() const { return nullptr; }
	// GaRunConfig::emit_runs as it ended up (a back end that builds the match words clears it when a row is such a character)
	virtual bool emittingRuns() const = 0;
	virtual int run() = 0;                                                   // device work only; returns ga_status
	// moves of job i: (*traceBytes)[outs[i].trace_off ..+trace_len); the bytes stay the backend's (valid until fetchDone or the next fetch)
	virtual int fetch(std::vector<GaJobOut>& outs, const uint8_t** traceBytes, uint64_t* nBytes) = 0;
	virtual void fetchDone() {}
	virtual GaRunStats stats() const = 0;
};

// returns nullptr + sets *status (GA_E_NO_DEVICE ...) on failure
GaBackendGraph* ga_backend_upload_graph(const GaFlatGraph& g, const GaHmmTables& hmm, int device, int* status);
// eq: per job and 64-row slice five 64-bit words (job j, slice s at (jobs[j].rows_off / 64 + s) * 5): the match words of the slice's rows
// against A, C, G, T and a meta word (bits 0-2 exact-compare code of the slice's last row, bit 3 = a row with an invalid character)
// rows: the row codes (one byte per padded read base) are only needed by the wave-per-read ladder kernels, so they are built and uploaded
// on demand: the provider returns them (building them at its first call)
typedef std::function<const std::vector<uint8_t>&()> GaRowsProvider;
// (eq or src: the finished words, or what a back end that builds them itself builds them from)
GaBackendBatch* ga_backend_create_batch(GaBackendGraph* g, GaRowsProvider rows, const uint64_t* eq, const GaEqSource* src, size_t eqWords, const std::vector<GaJob>& jobs,
                                        const GaRunConfig& cfg, int* status);

// the match words of ga_backend_create_batch from the row codes
inline void ga_build_eq_words(const uint8_t* rows, uint64_t nRows, uint64_t* eq)
{
	for (uint64_t s = 0; s < nRows / 64; s++)
	{
		uint64_t e[4] = {0, 0, 0, 0};
		uint32_t invalid = 0;
		const uint8_t* r = rows + s * 64;
#if defined(__SSE2__) && !defined(__HIP_DEVICE_COMPILE__)
		// bit b of 16 codes at a time: shifted up to the bytes' sign bits, collected by movemask
		__m128i any = _mm_setzero_si128();
		for (int q = 0; q < 4; q++)
		{
			const __m128i v = _mm_loadu_si128((const __m128i*)(r + 16 * q));
			any = _mm_or_si128(any, v);
			e[0] |= (uint64_t)(uint32_t)_mm_movemask_epi8(_mm_slli_epi16(v, 7)) << (16 * q);
			e[1] |= (uint64_t)(uint32_t)_mm_movemask_epi8(_mm_slli_epi16(v, 6)) << (16 * q);
			e[2] |= (uint64_t)(uint32_t)_mm_movemask_epi8(_mm_slli_epi16(v, 5)) << (16 * q);
			e[3] |= (uint64_t)(uint32_t)_mm_movemask_epi8(_mm_slli_epi16(v, 4)) << (16 * q);
		}
		invalid = (uint32_t)_mm_movemask_epi8(any);            // GA_ROW_INVALID is the codes' top bit
#else
		for (int i = 0; i < 64; i++)
		{
			const uint8_t c = r[i];
			for (int b = 0; b < 4; b++) e[b] |= (uint64_t)((c >> b) & 1) << i;
			invalid |= c & GA_ROW_INVALID;
		}
#endif
		uint64_t* o = eq + s * 5;
		o[0] = e[0]; o[1] = e[1]; o[2] = e[2]; o[3] = e[3];
		o[4] = (uint64_t)((r[63] >> 4) & 7) | (invalid ? 8u : 0u);
	}
}
