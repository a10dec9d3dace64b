// ga_backend.h -- the seam between the host library (graph model, job building, result
// assembly: ga_host.cpp) and whatever executes the extension program.  The product links
// ga_device.hip (gfx950, HIP).  tests/emul/ links a host emulation of the same program so the
// device logic can be checked in the CPU-only container; that object is never part of the
// shipped library.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "ga_types.h"

struct GaFlatGraph                      // host copy of what goes to HBM
{
	std::vector<uint64_t> node_start;   // n_nodes + 1
	std::vector<uint32_t> seq2;
	std::vector<uint32_t> in_off, in_nbr, out_off, out_nbr;
};

struct GaRunConfig
{
	int initial_bw = 0, ramp_bw = 0;
	uint32_t max_slices = 0;            // max over jobs of n_rows / 64
	uint32_t max_rows = 0;
};

struct GaRunStats
{
	double kernel_ms = 0;
	uint32_t slots = 0, waves_per_cu = 0;
	uint64_t scratch_bytes = 0;
	uint64_t jobs_retried = 0;
	uint64_t stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // diagnostic builds: summed shader cycles per phase
};

class GaBackendGraph { public: virtual ~GaBackendGraph() {} };
class GaBackendBatch
{
public:
	virtual ~GaBackendBatch() {}
	virtual int run() = 0;                                                   // device work only; returns ga_status
	virtual int fetch(std::vector<GaJobOut>& outs, std::vector<uint8_t>& traceBytes) = 0;     // moves of job i: traceBytes[outs[i].trace_off ..+trace_len)
	virtual GaRunStats stats() const = 0;
};

// returns nullptr + sets *status (GA_E_NO_DEVICE ...) on failure
GaBackendGraph* ga_backend_upload_graph(const GaFlatGraph& g, const GaHmmTables& hmm, int device, int* status);
GaBackendBatch* ga_backend_create_batch(GaBackendGraph* g, const std::vector<uint8_t>& rows, const std::vector<GaJob>& jobs,
                                        const GaRunConfig& cfg, int* status);
