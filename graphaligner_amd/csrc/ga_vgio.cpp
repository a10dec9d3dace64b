// ga_vgio.cpp -- the file formats on either side of the hot path, without libprotobuf:
//   * vg.Graph chunks -> graph             DirectedGraph::StreamVGGraphFromFile (BigraphToDigraph.cpp:106-135)
//   * seed GAM -> (read name, seed hit)     Aligner.cpp:253-271
//   * results -> GAM                        Aligner.cpp:173 (ids halved), 301-314 (one stream::write of all alignments)
// Framing is the reference's stream.hpp:24-118: a gzip stream of groups, each group = varint64
// count followed by count x (varint32 size, message).  Messages are proto3 (zero / empty fields
// are not written, vg.pb.cpp:3058-3087); field numbers from vg.pb.h:149-173, 262-284, 368-392,
// 478-490, 580-600, 684-702, 792-821, 906-960.  Only zlib is needed.
#include <zlib.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/graphaligner_amd.h"

namespace {

bool gunzipAll(const uint8_t* data, size_t len, std::vector<uint8_t>& out)
{
	// concatenated gzip members are allowed (one per stream::write call); zlib's counters are 32-bit, so input is fed in pieces
	size_t at = 0;
	while (at < len)
	{
		z_stream z;
		memset(&z, 0, sizeof(z));
		if (inflateInit2(&z, 16 + MAX_WBITS) != Z_OK) return false;
		size_t fed = at;                          // bytes of `data` handed to zlib so far
		z.next_in = const_cast<Bytef*>(data + at);
		z.avail_in = 0;
		int rc = Z_OK;
		uint8_t buf[1 << 16];
		while (rc != Z_STREAM_END)
		{
			if (z.avail_in == 0 && fed < len)
			{
				const size_t piece = std::min<size_t>(len - fed, 1u << 30);
				z.next_in = const_cast<Bytef*>(data + fed);
				z.avail_in = (uInt)piece;
				fed += piece;
			}
			z.next_out = buf;
			z.avail_out = sizeof(buf);
			rc = inflate(&z, Z_NO_FLUSH);
			if (rc != Z_OK && rc != Z_STREAM_END) { inflateEnd(&z); return false; }
			out.insert(out.end(), buf, buf + (sizeof(buf) - z.avail_out));
			if (rc == Z_OK && z.avail_in == 0 && fed >= len && z.avail_out != 0) { inflateEnd(&z); return false; }   // truncated
		}
		at = fed - z.avail_in;                    // where this member ended
		inflateEnd(&z);
	}
	return true;
}

bool gzipAll(const std::vector<uint8_t>& in, std::vector<uint8_t>& out)
{
	z_stream z;
	memset(&z, 0, sizeof(z));
	if (deflateInit2(&z, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 16 + MAX_WBITS, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
	out.clear();
	size_t fed = 0;
	uint8_t buf[1 << 16];
	int rc = Z_OK;
	while (rc != Z_STREAM_END)
	{
		if (z.avail_in == 0 && fed < in.size())
		{
			const size_t piece = std::min<size_t>(in.size() - fed, 1u << 30);
			z.next_in = const_cast<Bytef*>(in.data() + fed);
			z.avail_in = (uInt)piece;
			fed += piece;
		}
		z.next_out = buf;
		z.avail_out = sizeof(buf);
		rc = deflate(&z, fed >= in.size() ? Z_FINISH : Z_NO_FLUSH);
		if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) { deflateEnd(&z); return false; }
		out.insert(out.end(), buf, buf + (sizeof(buf) - z.avail_out));
	}
	deflateEnd(&z);
	return true;
}

struct Reader
{
	const uint8_t* p; const uint8_t* end; bool ok = true;
	Reader(const uint8_t* b, size_t n) : p(b), end(b + n) {}
	bool more() const { return ok && p < end; }
	uint64_t varint()
	{
		uint64_t r = 0; int s = 0;
		while (p < end) { uint8_t c = *p++; r |= (uint64_t)(c & 0x7f) << s; if (!(c & 0x80)) return r; s += 7; if (s > 63) break; }
		ok = false; return 0;
	}
	// next field: number, wire type; value in `v` (varint / fixed) or [data,len) for length-delimited
	bool field(int& num, int& wire, uint64_t& v, const uint8_t*& data, size_t& len)
	{
		if (!more()) return false;
		uint64_t key = varint();
		num = (int)(key >> 3); wire = (int)(key & 7);
		data = nullptr; len = 0; v = 0;
		if (wire == 0) v = varint();
		else if (wire == 2) { len = (size_t)varint(); if (!ok || len > (size_t)(end - p)) { ok = false; return false; } data = p; p += len; }
		else if (wire == 1) { if (end - p < 8) { ok = false; return false; } memcpy(&v, p, 8); p += 8; }
		else if (wire == 5) { if (end - p < 4) { ok = false; return false; } uint32_t t; memcpy(&t, p, 4); v = t; p += 4; }
		else { ok = false; return false; }
		return ok;
	}
};

template <typename F> bool forEachMessage(const std::vector<uint8_t>& raw, F f)
{
	Reader r(raw.data(), raw.size());
	while (r.more())
	{
		uint64_t count = r.varint();
		if (!r.ok) return false;
		for (uint64_t i = 0; i < count; i++)
		{
			size_t n = (size_t)r.varint();
			if (!r.ok || n > (size_t)(r.end - r.p)) return false;
			if (!f(r.p, n)) return false;
			r.p += n;
		}
	}
	return true;
}

struct Writer
{
	std::vector<uint8_t> b;
	void varint(uint64_t v) { while (v >= 0x80) { b.push_back((uint8_t)(v | 0x80)); v >>= 7; } b.push_back((uint8_t)v); }
	void key(int num, int wire) { varint((uint64_t)num << 3 | wire); }
	void intField(int num, int64_t v) { if (v != 0) { key(num, 0); varint((uint64_t)v); } }           // proto3: zeros are not written
	void bytesField(int num, const void* d, size_t n, bool always = false) { if (n || always) { key(num, 2); varint(n); b.insert(b.end(), (const uint8_t*)d, (const uint8_t*)d + n); } }
	void message(int num, const Writer& w) { key(num, 2); varint(w.b.size()); b.insert(b.end(), w.b.begin(), w.b.end()); }
};

}  // namespace

extern "C" {

int ga_graph_load_vg(ga_graph_t* g, const void* bytes, size_t len)
{
	if (!g || !bytes) return GA_E_INVALID;
	std::vector<uint8_t> raw;
	if (!gunzipAll((const uint8_t*)bytes, len, raw)) return GA_E_INVALID;
	// pass 1: nodes of every chunk, pass 2: edges (BigraphToDigraph.cpp:109-133)
	int status = GA_S_OK;
	for (int pass = 0; pass < 2 && status == GA_S_OK; pass++)
	{
		bool ok = forEachMessage(raw, [&](const uint8_t* m, size_t n) {
			Reader gr(m, n);
			int num, wire; uint64_t v; const uint8_t* d; size_t dl;
			while (gr.field(num, wire, v, d, dl))
			{
				if (pass == 0 && num == 1 && wire == 2)
				{
					Reader nr(d, dl);
					std::string seq; int64_t id = 0;
					int fn, fw; uint64_t fv; const uint8_t* fd; size_t fl;
					while (nr.field(fn, fw, fv, fd, fl)) { if (fn == 1 && fw == 2) seq.assign((const char*)fd, fl); else if (fn == 3 && fw == 0) id = (int64_t)fv; }
					if (!nr.ok) return false;
					status = ga_graph_add_bigraph_node(g, id, seq.data(), seq.size());
					if (status) return false;
				}
				else if (pass == 1 && num == 2 && wire == 2)
				{
					Reader er(d, dl);
					int64_t from = 0, to = 0; int fromStart = 0, toEnd = 0;
					int fn, fw; uint64_t fv; const uint8_t* fd; size_t fl;
					while (er.field(fn, fw, fv, fd, fl))
					{
						if (fw != 0) continue;
						if (fn == 1) from = (int64_t)fv; else if (fn == 2) to = (int64_t)fv; else if (fn == 3) fromStart = fv != 0; else if (fn == 4) toEnd = fv != 0;
					}
					if (!er.ok) return false;
					status = ga_graph_add_bigraph_edge(g, from, fromStart, to, toEnd);
					if (status) return false;
				}
			}
			return gr.ok;
		});
		if (!ok && status == GA_S_OK) status = GA_E_INVALID;
	}
	if (status) return status;
	return ga_graph_finalize(g, 0);            // the vg loader leaves DBGOverlap at 0 (AlignmentGraph.cpp:13)
}

int ga_gam_decode_seeds(const void* bytes, size_t len, ga_named_seed_t** out, size_t* nOut)
{
	if (!bytes || !out || !nOut) return GA_E_INVALID;
	std::vector<uint8_t> raw;
	if (!gunzipAll((const uint8_t*)bytes, len, raw)) return GA_E_INVALID;
	std::vector<ga_named_seed_t> seeds;
	std::vector<std::string> names;
	bool ok = forEachMessage(raw, [&](const uint8_t* m, size_t n) {
		// seedhit.path().mapping(0).position().node_id(), query_position(), position().is_reverse()  (Aligner.cpp:269)
		Reader ar(m, n);
		ga_named_seed_t s;
		memset(&s, 0, sizeof(s));
		std::string name;
		bool firstMapping = true;
		int num, wire; uint64_t v; const uint8_t* d; size_t dl;
		while (ar.field(num, wire, v, d, dl))
		{
			if (num == 3 && wire == 2) name.assign((const char*)d, dl);
			else if (num == 7 && wire == 0) s.seed.read_pos = v;
			else if (num == 2 && wire == 2)
			{
				Reader pr(d, dl);
				int pn, pw; uint64_t pv; const uint8_t* pd; size_t pl;
				while (pr.field(pn, pw, pv, pd, pl))
				{
					if (pn != 2 || pw != 2 || !firstMapping) continue;
					firstMapping = false;
					Reader mr(pd, pl);
					int mn, mw; uint64_t mv; const uint8_t* md; size_t ml;
					while (mr.field(mn, mw, mv, md, ml))
					{
						if (mn != 1 || mw != 2) continue;
						Reader qr(md, ml);
						int qn, qw; uint64_t qv; const uint8_t* qd; size_t ql;
						while (qr.field(qn, qw, qv, qd, ql)) { if (qw != 0) continue; if (qn == 1) s.seed.node_id = (int64_t)qv; else if (qn == 4) s.seed.reverse = qv != 0; }
					}
				}
			}
		}
		if (!ar.ok) return false;
		names.push_back(name);
		seeds.push_back(s);
		return true;
	});
	if (!ok) return GA_E_INVALID;
	size_t bytesNeeded = seeds.size() * sizeof(ga_named_seed_t);
	for (auto& nm : names) bytesNeeded += nm.size() + 1;
	char* block = (char*)malloc(bytesNeeded + 1);
	if (!block) return GA_E_INVALID;
	ga_named_seed_t* arr = (ga_named_seed_t*)block;
	char* str = block + seeds.size() * sizeof(ga_named_seed_t);
	for (size_t i = 0; i < seeds.size(); i++)
	{
		arr[i] = seeds[i];
		memcpy(str, names[i].c_str(), names[i].size() + 1);
		arr[i].read_name = str;
		str += names[i].size() + 1;
	}
	*out = arr;
	*nOut = seeds.size();
	return GA_S_OK;
}

int ga_results_encode_gam(const ga_results_t* r, const ga_read_t* reads, int halveNodeIds, void** out, size_t* outLen)
{
	if (!r || !reads || !out || !outLen) return GA_E_INVALID;
	Writer group;
	uint64_t count = 0;
	Writer body;
	for (size_t i = 0; i < r->n_reads; i++)
	{
		const ga_read_result_t& rr = r->reads[i];
		if (rr.failed || rr.status != GA_S_OK) continue;                          // failed alignments are not output (Aligner.cpp:153-164)
		Writer path;
		for (uint64_t k = rr.first_mapping; k < rr.first_mapping + rr.n_mappings; k++)
		{
			const ga_mapping_t& m = r->mappings[k];
			Writer pos;
			pos.intField(1, halveNodeIds ? m.node_id / 2 : m.node_id);           // replaceDigraphNodeIdsWithOriginalNodeIds (Aligner.cpp:83-91)
			pos.intField(2, m.offset);
			pos.intField(4, m.is_reverse);
			Writer edit;
			edit.intField(1, m.from_length);
			edit.intField(2, m.to_length);
			edit.bytesField(3, r->edit_bytes + m.edit_seq_off, (size_t)m.to_length);
			Writer mapping;
			mapping.message(1, pos);
			mapping.message(2, edit);
			mapping.intField(5, m.rank);
			path.message(2, mapping);
		}
		Writer aln;
		aln.bytesField(1, reads[i].sequence, reads[i].length);
		aln.message(2, path);
		aln.bytesField(3, reads[i].name, reads[i].name ? strlen(reads[i].name) : 0);
		aln.intField(6, rr.score);
		aln.intField(7, (int64_t)rr.query_position);
		body.varint(aln.b.size());
		body.b.insert(body.b.end(), aln.b.begin(), aln.b.end());
		count++;
	}
	group.varint(count);
	group.b.insert(group.b.end(), body.b.begin(), body.b.end());
	std::vector<uint8_t> z;
	if (!gzipAll(group.b, z)) return GA_E_INVALID;
	void* p = malloc(z.size() ? z.size() : 1);
	if (!p) return GA_E_INVALID;
	memcpy(p, z.data(), z.size());
	*out = p;
	*outLen = z.size();
	return GA_S_OK;
}

void ga_bytes_free(void* p) { free(p); }

}  // extern "C"
