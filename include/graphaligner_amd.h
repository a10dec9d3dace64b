/* graphaligner_amd.h -- C ABI of the MI355X-native seed-and-extend aligner.
 *
 * Drop-in boundary: the two free functions the reference's driver calls per read,
 *     AlignmentResult AlignOneWay(const AlignmentGraph&, const std::string& seq_id,
 *                                 const std::string& sequence, int initialBandwidth,
 *                                 int rampBandwidth, size_t dynamicRowStart,
 *                                 const std::vector<std::tuple<int,size_t,bool>>& seedHits);
 * (reference GraphAlignerWrapper.h:53-54, called from Aligner.cpp:128,140), plus the graph
 * construction calls its loaders make (AlignmentGraph.h:25-28, BigraphToDigraph.cpp:106-189).
 * A GPU cannot be fed one read per call, so the ABI is batch-first; AlignOneWay is the
 * one-element case (see INTEGRATION.md for the C++ shim a maintainer would add).
 *
 * Plain pointers and sizes only.  No exceptions cross this boundary: every function returns
 * a ga_status, and per-read failures are reported in ga_read_result.status / .failed
 * (replacing ThreadReadAssertion::AssertionFailure, GraphAligner.h assert()s, and the
 * alignmentFailed / INT32_MAX-score convention of GraphAligner.h:636-641).
 *
 * The library needs a gfx950 device: ga_graph_upload / ga_batch_run return GA_E_NO_DEVICE
 * when none is usable.  There is no CPU fallback.
 *
 * Memory the library keeps between calls: a scratch pool, a pinned download buffer, idle device blocks of finished batches and up to
 * six pinned blocks for the batches' copies of the reads per uploaded graph (freed with the graph; a batch's copy of its reads is also
 * what its results' edit_bytes point into, so such a block lives until both the batch and its results are freed), and up to
 * GA_RESULT_POOL_MB (default 4096) of recycled result arrays per process.  Environment knobs, none of them needed:
 * GA_HOST_THREADS (host threads for job building and result assembly; default: the CPUs the process may use), GA_RESULT_POOL_MB,
 * GA_LANES=1/0 (force / forbid the lanes = reads kernel as the first pass; default by the graph's mean node length),
 * GA_LANES_SPREAD=0 / k (full waves / k reads per wave instead of spreading a small batch over all wave slots), GA_DEBUG_PASSES / GA_DEBUG_COLLECT
 * (one line per kernel pass / per host stage on stderr).
 */
#ifndef GRAPHALIGNER_AMD_H
#define GRAPHALIGNER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum ga_status {
	GA_S_OK = 0,
	/* per-read outcomes (ga_read_result.status) */
	GA_S_ASSERTION = 1,         /* the reference's always-on assert() throws for this read (Aligner.cpp:143) */
	GA_S_UNSUPPORTED_BAND = 2,  /* internal: a band of >= 200000 bp, where the reference uses its sparse method (GraphAligner.h:2483); resolved by the
	                               last kernel variant, which carries that method and the backtrace override -- returned only when that pass
	                               could not get its device memory */
	GA_S_BAD_SEED = 3,          /* seed node id unknown: std::out_of_range in the reference (GraphAligner.h:423) */
	GA_S_CAPACITY = 10,         /* device buffers too small even after the automatic retry */
	GA_S_UNSUPPORTED_CYCLE = 20,/* internal: band subgraph has a cycle (GraphAligner.h:2362-2397); resolved by the general kernel variants, not returned */
	GA_S_UNSUPPORTED_RAMP = 21, /* internal: ramp redo (GraphAligner.h:2648-2719) taken; resolved by the general kernel variants, not returned */
	/* call-level errors */
	GA_E_INVALID = 100,
	GA_E_NO_DEVICE = 101,
	GA_E_DEVICE = 102,
	GA_E_NOT_FINALIZED = 103
} ga_status;

/* ---- graph: mirrors AlignmentGraph's public build interface (AlignmentGraph.h:25-28) ---------- */
typedef struct ga_graph ga_graph_t;

ga_graph_t* ga_graph_create(void);                                         /* AlignmentGraph::AlignmentGraph (AlignmentGraph.cpp:12-31) */
void ga_graph_destroy(ga_graph_t* g);
/* AlignmentGraph::AddNode (AlignmentGraph.cpp:47-89): digraph id, ACGT sequence; duplicate ids are ignored */
int ga_graph_add_node(ga_graph_t* g, int64_t digraph_id, const char* seq, size_t len, int reverse_node);
/* AlignmentGraph::AddEdgeNodeId (AlignmentGraph.cpp:91-106): duplicate edges are ignored */
int ga_graph_add_edge(ga_graph_t* g, int64_t from_digraph_id, int64_t to_digraph_id);
/* A FINISHED graph mirrored verbatim: the in- and out-neighbour lists of one node exactly as AlignmentGraph holds them
 * (inNeighbors[i], outNeighbors[i], AlignmentGraph.h:49-50), digraph ids, before ga_graph_finalize.  Both orders are part of the result
 * (traceback ties follow inNeighbors, GraphAligner.h:503; the band's heap is fed in outNeighbors order, :1132-1134, 1153-1155), and a
 * replay of edges grouped by target cannot reproduce both: use this instead of ga_graph_add_edge when copying a built AlignmentGraph.
 * A node given lists here ignores edges added for it with ga_graph_add_edge. */
int ga_graph_set_neighbors(ga_graph_t* g, int64_t digraph_id, const int64_t* in_ids, size_t n_in, const int64_t* out_ids, size_t n_out);
/* the loaders' bidirected -> directed conversion (BigraphToDigraph.cpp:27-56, 58-104):
 * node id -> 2*id (forward) and 2*id+1 (reverse complement); each edge -> two directed edges */
int ga_graph_add_bigraph_node(ga_graph_t* g, int64_t id, const char* seq, size_t len);
int ga_graph_add_bigraph_edge(ga_graph_t* g, int64_t from, int from_start, int64_t to, int to_end);
/* AlignmentGraph::Finalize (AlignmentGraph.cpp:108-154); dbg_overlap = AlignmentGraph::DBGOverlap */
int ga_graph_finalize(ga_graph_t* g, int dbg_overlap);
/* DirectedGraph::StreamGFAGraphFromFile (BigraphToDigraph.cpp:137-189) over an in-memory GFA text; finalizes */
int ga_graph_load_gfa(ga_graph_t* g, const char* text, size_t len);
/* the same with every segment longer than max_node_len cut into a chain of pieces (the first keeps the segment's id, the others get new
 * ids above the largest id of the file): the reference never splits nodes and leaves bands of >= 200 000 bp to its sparse method
 * (GraphAlignerCommon.h:10), so a graph of long segments (a single-contig GFA) only reaches the bit-vector path when cut up.  Blunt
 * graphs only (overlap 0).  This is an OPTION, not the reference's behaviour: alignments are those of the cut graph. */
int ga_graph_load_gfa_split(ga_graph_t* g, const char* text, size_t len, uint32_t max_node_len);
/* where a (bigraph) node id of a split graph comes from: 0 and (original id, start inside it, its length), or 1 when the id is not a piece */
int ga_graph_split_lookup(const ga_graph_t* g, int64_t bigraph_id, int64_t* orig_id, uint64_t* start, uint64_t* orig_len);
/* copy the flattened graph into the HBM of `device` (replicated per GPU; one process per GPU) */
int ga_graph_upload(ga_graph_t* g, int device);
int64_t ga_graph_node_count(const ga_graph_t* g);   /* AlignmentGraph::NodeSize, including the two dummy nodes */
int64_t ga_graph_bp(const ga_graph_t* g);           /* AlignmentGraph::SizeInBp */

/* ---- reads and seeds ----------------------------------------------------------------------------- */
typedef struct ga_read {
	const char* name;        /* seq_id */
	const char* sequence;
	size_t length;
} ga_read_t;

typedef struct ga_seed {     /* std::tuple<int,size_t,bool> seedHits (Aligner.cpp:269) */
	int64_t node_id;         /* BIGRAPH node id */
	uint64_t read_pos;       /* query_position */
	int32_t reverse;
	int32_t reserved;
} ga_seed_t;

/* ---- results: flat POD mirror of AlignmentResult (GraphAlignerWrapper.h:10-51) ---------------------- */
typedef struct ga_mapping {  /* vg::Mapping with its single vg::Edit (GraphAligner.h:782-847) */
	int64_t node_id;         /* DIGRAPH id, as the engine returns it; the driver halves it (Aligner.cpp:83-91) */
	int32_t is_reverse;
	int32_t rank;
	int64_t offset;          /* Position.offset: set on the first mapping only */
	int64_t from_length;
	int64_t to_length;
	uint64_t edit_seq_off;   /* Edit.sequence = results->edit_bytes[edit_seq_off .. +to_length) */
} ga_mapping_t;

typedef struct ga_trace_item {   /* AlignmentResult::TraceItem (GraphAlignerWrapper.h:22-31) */
	int32_t node_id;
	int32_t reverse;
	uint64_t offset;
	uint64_t read_pos;
	int32_t type;            /* 1 MATCH 2 MISMATCH 3 INSERTION 4 DELETION 5 FORWARDBACKWARDSPLIT */
	char graph_char;
	char read_char;
	char pad[2];
} ga_trace_item_t;

typedef struct ga_read_result {
	int32_t status;          /* ga_status, per read */
	int32_t failed;          /* AlignmentResult::alignmentFailed */
	int32_t score;           /* vg::Alignment.score; INT32_MAX when failed */
	int32_t reserved;        /* diagnostic: 0 when the first kernel pass finished all of the read's extensions, else the number of the last pass that did */
	uint64_t alignment_start, alignment_end, query_position;
	uint64_t first_mapping, n_mappings;     /* into results->mappings */
	uint64_t first_trace, n_trace;          /* into results->trace (empty unless requested) */
	uint64_t column_updates;                /* band columns computed for this read (first pass) */
} ga_read_result_t;

/* The three arrays behind `reads` are written by several host threads at places fixed before the lengths are known, so they have GAPS:
 * n_mappings / n_edit_bytes / n_trace are the arrays' extents, not counts of valid entries, and the bytes between one read's entries
 * and the next's are unspecified (recycled memory).  Valid entries are exactly those a read's record points at:
 * mappings[first_mapping .. first_mapping + n_mappings), trace[first_trace .. first_trace + n_trace), and per mapping
 * edit_bytes[edit_seq_off .. edit_seq_off + to_length).  Never iterate an array from 0 to its extent. */
typedef struct ga_results {
	size_t n_reads;
	const ga_read_result_t* reads;
	size_t n_mappings;                      /* extent of `mappings` (see above) */
	const ga_mapping_t* mappings;
	size_t n_edit_bytes;                    /* extent of `edit_bytes` */
	const char* edit_bytes;
	size_t n_trace;                         /* extent of `trace` */
	const ga_trace_item_t* trace;
} ga_results_t;

/* ---- alignment ----------------------------------------------------------------------------------------- */
/* reads[i] owns seeds[seed_offsets[i] .. seed_offsets[i+1]).  flags: GA_F_TRACE fills TraceItem lists. */
#define GA_F_TRACE 1u
int ga_align_batch(const ga_graph_t* g, const ga_read_t* reads, size_t n_reads, const ga_seed_t* seeds, const size_t* seed_offsets,
                   int initial_bandwidth, int ramp_bandwidth, uint32_t flags, ga_results_t** out);
void ga_results_free(ga_results_t* r);
/* results over a split graph (ga_graph_load_gfa_split) on the nodes of the GFA file: runs of pieces merged, offsets and trace items remapped */
int ga_results_unsplit(const ga_graph_t* g, const ga_results_t* in, ga_results_t** out);

/* staged form of the same call, for callers that keep inputs resident in HBM and for measurement:
 *   prepare = validate seeds, build extension jobs, upload reads;  run = the device work only;
 *   collect = download + assemble AlignmentResults. */
/* Threading: a graph is read-only once uploaded and may serve any number of batches from any threads.  One batch is used by one thread
 * at a time (prepare -> run -> collect may run on three different threads, as the overlapped pipeline does, but never concurrently on
 * the same batch).  Batches of one graph that RUN concurrently do not share scratch: the first takes the graph's pool, the others
 * allocate their own for the duration. */
typedef struct ga_batch ga_batch_t;
int ga_batch_prepare(const ga_graph_t* g, const ga_read_t* reads, size_t n_reads, const ga_seed_t* seeds, const size_t* seed_offsets,
                     int initial_bandwidth, int ramp_bandwidth, uint32_t flags, ga_batch_t** out);
int ga_batch_run(ga_batch_t* b);
int ga_batch_collect(ga_batch_t* b, ga_results_t** out);
void ga_batch_free(ga_batch_t* b);

typedef struct ga_batch_stats {
	uint64_t n_jobs;             /* extension jobs (read directions) */
	uint64_t column_updates;     /* sum over jobs of band columns computed (unit of work) */
	uint64_t slices;             /* 64-row slices computed */
	uint64_t jobs_retried;       /* jobs the first pass handed to the wave-per-read kernel ladder */
	double kernel_ms;            /* HIP-event time of the extension kernel(s) of the last ga_batch_run, all passes */
	double prep_kernel_ms;       /* HIP-event time of the read-coding kernel */
	uint32_t slots, waves_per_cu;
	uint64_t scratch_bytes;
	uint64_t stamps[8];          /* diagnostic builds (GA_STAMPS) only: shader cycles per phase, summed over jobs */
	double main_kernel_ms;       /* HIP-event time of the first pass alone (the lanes = reads kernel over all jobs) */
	int32_t main_variant;        /* > 0: lanes = reads kernel, band nodes per lane * 1000 + record block * 10 + (1 when 32 lanes per wave);
	                                < 0: wave-per-read kernel with that many band nodes in LDS */
	int32_t reserved;            /* 1: the batch's results need no cell lists (no GA_F_TRACE, one seed at the first base of every read, IUPAC
	                                characters only) and the first-pass kernel handed back node runs instead of one byte per move */
} ga_batch_stats_t;
int ga_batch_stats(const ga_batch_t* b, ga_batch_stats_t* out);

/* ---- file formats either side of the path (no libprotobuf; zlib only) --------------------------------- */
/* gzip-framed vg.Graph chunks (stream.hpp:24-118) -> graph, as DirectedGraph::StreamVGGraphFromFile
 * (BigraphToDigraph.cpp:106-135): all nodes first, then all edges, then Finalize; DBGOverlap stays 0 */
int ga_graph_load_vg(ga_graph_t* g, const void* bytes, size_t len);
/* seed GAM -> (read name, seed hit): mapping(0).position().node_id / is_reverse, query_position (Aligner.cpp:253-271) */
typedef struct ga_named_seed { const char* read_name; ga_seed_t seed; } ga_named_seed_t;
int ga_gam_decode_seeds(const void* bytes, size_t len, ga_named_seed_t** out, size_t* n);   /* free with ga_bytes_free(*out) */
/* results -> one GAM group of vg.Alignment (Aligner.cpp:301-314); failed reads are skipped (Aligner.cpp:153-164);
 * halve_node_ids applies replaceDigraphNodeIdsWithOriginalNodeIds (Aligner.cpp:83-91) */
int ga_results_encode_gam(const ga_results_t* r, const ga_read_t* reads, int halve_node_ids, void** out, size_t* out_len);
void ga_bytes_free(void* p);

const char* ga_status_string(int status);
const char* ga_version(void);

#ifdef __cplusplus
}
#endif
#endif
