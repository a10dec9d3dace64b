// ga_types.h -- plain structs shared by the host library and the device program.
#pragma once
#include <stdint.h>

// per-job status codes (device) -- mirrored in include/graphaligner_amd.h
enum GaStatus : int32_t {
	GA_OK = 0,
	GA_ASSERTION = 1,          // an always-on assert() of the reference would have thrown for this read
	GA_UNSUPPORTED_BAND = 2,   // band >= 200000 bp: the reference switches to its sparse method (GraphAligner.h:2483); internal: the job moves to the
	                           // last kernel of the ladder, which carries that method (ga_sparse.h)
	GA_BAD_SEED = 3,           // seed node id not in the graph (std::out_of_range in the reference, GraphAligner.h:423)
	GA_CAP_NODES = 10,         // band holds more nodes than this kernel variant keeps in LDS -> rerun with the wide variant
	GA_CAP_COLS = 11,          // band holds more columns than the slot's end-score buffers
	GA_CAP_ARENA = 12,         // slot arena (stored VP/VN words) exhausted
	GA_CAP_TRACE = 13,         // trace output buffer exhausted
	GA_CAP_HEAP = 14,          // band-projection heap exhausted
	GA_UNSUPPORTED_CYCLE = 20, // band subgraph has a cycle (iterative row confirmation, GraphAligner.h:2362-2397, not built on device)
	GA_UNSUPPORTED_RAMP = 21,  // the ramp-redo path (GraphAligner.h:2648-2719) would have been taken
	GA_PUNT = 22,              // the lanes = reads kernel declines the job (short read, node degree > 4, ...): the wave-per-read ladder runs it
	GA_NOT_RUN = 99,
};

struct GaDevGraph {
	uint32_t n_nodes;            // including the two dummy nodes
	uint32_t reserved;
	const uint64_t* node_start;  // [n_nodes + 1] first column of each node; [n_nodes] = total bp
	const uint32_t* seq2;        // 2 bits per column: A=0 C=1 G=2 T=3 (dummy columns hold 0)
	const uint32_t* in_off;      // [n_nodes + 1]
	const uint32_t* in_nbr;      // in-neighbours in insertion order (AlignmentGraph.cpp:104)
	const uint32_t* out_off;     // [n_nodes + 1]
	const uint32_t* out_nbr;     // out-neighbours in insertion order (AlignmentGraph.cpp:105)
	// one 64-byte record per node so that everything a band step needs about a node arrives with one request:
	// [0..1] first column, [2] length, [3] in-degree | out-degree << 16 (capped at 0xffff),
	// [4..7] first four out-neighbours, [8..11] first four in-neighbours, [12..15] the lengths of those in-neighbours
	const uint32_t* node_rec;
};
#define GA_NODE_REC_WORDS 16

// log-space Viterbi constants, computed on the host with the same libm calls in the same
// order as AlignmentCorrectnessEstimation.cpp:6-36,81-83; the device only adds and compares.
struct GaHmmTables {
	double init_correct, init_wrong;
	double c2c, f2c, c2f, f2f;
	double correct_mult[65], wrong_mult[65];
};

// one extension job = one direction of one (read, seed)
struct GaJob {
	uint64_t rows_off;     // offset of this job's row codes in the rows buffer
	uint32_t n_rows;       // padded to a multiple of 64
	uint32_t seed_node;    // graph node index the extension starts in
	uint32_t trace_rows;   // rows of the trace that count: cells at rows >= this are dropped from its end (GraphAligner.h:3051-3055, 3069-3073)
	uint32_t reserved;
};

struct GaJobOut {
	int32_t status;
	int32_t score;         // min score of the last kept slice (INT32_MAX when nothing was kept)
	uint32_t n_valid;      // slices kept after trimming (= bandwidthPerSlice.size())
	uint32_t n_run;        // slices computed in the first pass
	uint32_t trace_len;    // backward moves written (one byte each), from the start cell down to row 0
	uint32_t max_band_nodes;
	uint64_t n_columns;    // column updates = sum over computed slices of band columns
	uint64_t trace_off;    // byte offset of this job's moves inside the trace pool
	uint32_t start_node, start_offset, start_row, reserved2;   // where the traceback starts (last kept slice, last row)
	uint32_t n_node_steps;  // moves that left a node through its first column: the path has at most this many + 1 node runs
	uint32_t reserved3;     // 1: the job's bytes in the trace pool are node runs (5 words each: node, first offset, first row, last offset,
	                        // last row; from the read's end to its start; trace_len = their number) instead of one byte per move
	uint64_t stamps[8];    // diagnostic builds only (GA_STAMPS): shader cycles per phase; zero otherwise
};

// one traceback move, one byte: bits 0-1 = 0 left (same row), 1 diagonal, 2 up (same column);
// bits 2-7 = which in-neighbour (insertion order) the move enters when it leaves a node's first column
enum { GA_MOVE_LEFT = 0, GA_MOVE_DIAG = 1, GA_MOVE_UP = 2 };

struct GaLaunch {
	GaDevGraph graph;
	const GaHmmTables* hmm;
	const uint8_t* rows;        // row codes: bits 0-3 match mask over A,C,G,T; bits 4-6 exact code (7 = none); bit 7 invalid char
	const GaJob* jobs;
	GaJobOut* outs;
	uint8_t* traces;            // trace pool (bytes); each finished job claims its moves, rounded up to 4 bytes
	uint64_t* trace_top;        // device bump counter over the pool (bytes)
	uint64_t trace_pool_cap;    // bytes
	const uint32_t* job_list;   // optional indirection (retry launches): job index = job_list[k]; nullptr = identity
	uint32_t* next_job;         // device work counter
	uint8_t* scratch;           // [n_slots][slot_bytes]
	uint64_t slot_bytes;
	uint32_t n_jobs;
	uint32_t trace_cap;         // per-slot staging capacity (moves = bytes)
	uint32_t cap_cols;          // end-score buffer capacity per slice (columns)
	uint32_t max_slices;        // per job
	uint64_t arena_words;       // u32 words of slice storage per slot
	int32_t initial_bw, ramp_bw;
	uint32_t sparse_bw;         // the kernel variant with the sparse method: the bandwidth its per-slot tables are laid out for (else 0)
	uint32_t reserved;
};

// row code helpers (host side builds them; GraphAligner.h:2039-2110 for the match sets)
#define GA_ROW_INVALID 0x80
