// ga_wave_emul.h -- TEST-ONLY host back end of the wave64 vocabulary of graphaligner_amd/csrc/ga_wave.h: a per-lane value is an array
// of 64 and every primitive is a loop.  It exists so that the exact device program (ga_kernel.h, ga_sparse.h) can be checked against
// the oracle in the CPU-only build container; the test build (tests/emul/Makefile) substitutes it for ga_wave.h through
// -DGA_WAVE_HEADER.  Never part of the product library.
#pragma once
#include <stdint.h>

namespace gaw {

constexpr int LANES = 64;
constexpr int INF = 0x3fffffff;

// =========================================================================================
// host emulation
// =========================================================================================
#define GA_FN inline
#define GA_LANE0 true

struct VB { bool v[LANES]; };
struct VI
{
	int v[LANES];
	VI() {}
	VI(int s) { for (int i = 0; i < LANES; i++) v[i] = s; }
};
struct VU
{
	uint64_t v[LANES];
	VU() {}
	VU(uint64_t s) { for (int i = 0; i < LANES; i++) v[i] = s; }
};

#define GAW_BIN(OP) \
	inline VI operator OP(const VI& a, const VI& b) { VI r; for (int i = 0; i < LANES; i++) r.v[i] = a.v[i] OP b.v[i]; return r; } \
	inline VI operator OP(const VI& a, int b) { VI r; for (int i = 0; i < LANES; i++) r.v[i] = a.v[i] OP b; return r; } \
	inline VI operator OP(int a, const VI& b) { VI r; for (int i = 0; i < LANES; i++) r.v[i] = a OP b.v[i]; return r; }
GAW_BIN(+) GAW_BIN(-) GAW_BIN(&) GAW_BIN(|) GAW_BIN(>>) GAW_BIN(<<) GAW_BIN(*)
#undef GAW_BIN
#define GAW_CMP(OP) \
	inline VB operator OP(const VI& a, const VI& b) { VB r; for (int i = 0; i < LANES; i++) r.v[i] = a.v[i] OP b.v[i]; return r; } \
	inline VB operator OP(const VI& a, int b) { VB r; for (int i = 0; i < LANES; i++) r.v[i] = a.v[i] OP b; return r; }
GAW_CMP(==) GAW_CMP(<) GAW_CMP(>) GAW_CMP(!=)
#undef GAW_CMP
inline VB operator&&(const VB& a, const VB& b) { VB r; for (int i = 0; i < LANES; i++) r.v[i] = a.v[i] && b.v[i]; return r; }
inline VB operator&&(const VB& a, bool b) { VB r; for (int i = 0; i < LANES; i++) r.v[i] = a.v[i] && b; return r; }
inline VU operator&(uint64_t a, const VU& b) { VU r; for (int i = 0; i < LANES; i++) r.v[i] = a & b.v[i]; return r; }

inline VU operator&(const VU& a, const VU& b) { VU r; for (int i = 0; i < LANES; i++) r.v[i] = a.v[i] & b.v[i]; return r; }
// lane i receives lane i+1; lane 63 receives `fill`
inline VI shl1(const VI& x, int fill) { VI r; r.v[LANES - 1] = fill; for (int i = 0; i + 1 < LANES; i++) r.v[i] = x.v[i + 1]; return r; }
// lane i receives x[idx_i] (idx taken modulo 64)
// every lane receives x[lane]; the device form goes through the LDS crossbar (ds_bpermute), not the VALU
inline VI lane_broadcast(const VI& x, int lane) { VI r; for (int i = 0; i < LANES; i++) r.v[i] = x.v[lane & 63]; return r; }
inline VI shr1v(const VI& x, const VI& fill) { VI r; r.v[0] = fill.v[0]; for (int i = 1; i < LANES; i++) r.v[i] = x.v[i - 1]; return r; }
inline VI bit_extract_v(const VI& x, const VI& bit) { VI r; for (int i = 0; i < LANES; i++) r.v[i] = (x.v[i] >> bit.v[i]) & 1; return r; }
inline VI lane_gather(const VI& x, const VI& idx) { VI r; for (int i = 0; i < LANES; i++) r.v[i] = x.v[idx.v[i] & 63]; return r; }
// bit `pos` of a per-lane 64-bit word (0 when pos is outside 0..63)
inline VI bit64_at(const VU& w, const VI& pos) { VI r; for (int i = 0; i < LANES; i++) r.v[i] = (pos.v[i] < 0 || pos.v[i] > 63) ? 0 : (int)((w.v[i] >> pos.v[i]) & 1); return r; }
// per-lane word with the low `nbits` bits set (nbits <= 0 -> 0, >= 64 -> all)
inline VU mask_low_bits(const VI& nbits) { VU r; for (int i = 0; i < LANES; i++) r.v[i] = nbits.v[i] <= 0 ? 0ull : nbits.v[i] >= 64 ? ~0ull : ((1ull << nbits.v[i]) - 1); return r; }
inline VI lane_iota() { VI r; for (int i = 0; i < LANES; i++) r.v[i] = i; return r; }
inline VI vmin(const VI& a, const VI& b) { VI r; for (int i = 0; i < LANES; i++) r.v[i] = a.v[i] < b.v[i] ? a.v[i] : b.v[i]; return r; }
inline VI vmin(const VI& a, int b) { return vmin(a, VI(b)); }
inline VI select(const VB& c, const VI& a, const VI& b) { VI r; for (int i = 0; i < LANES; i++) r.v[i] = c.v[i] ? a.v[i] : b.v[i]; return r; }
inline VU select(const VB& c, const VU& a, const VU& b) { VU r; for (int i = 0; i < LANES; i++) r.v[i] = c.v[i] ? a.v[i] : b.v[i]; return r; }
inline VI bit_extract(const VI& x, int bit) { VI r; for (int i = 0; i < LANES; i++) r.v[i] = (x.v[i] >> bit) & 1; return r; }
// x mod B for a compile-time B (x >= 0)
template <int B> inline VI vmod(const VI& x) { VI r; for (int i = 0; i < LANES; i++) r.v[i] = (int)((unsigned)x.v[i] % (unsigned)B); return r; }
inline VI operator|(const VI& a, bool b) { VI r; for (int i = 0; i < LANES; i++) r.v[i] = a.v[i] | (b ? 1 : 0); return r; }
inline VI vpopc(const VU& a) { VI r; for (int i = 0; i < LANES; i++) r.v[i] = __builtin_popcountll(a.v[i]); return r; }
// bits 0..lane of a 64-bit word
inline VU low_mask_through_lane() { VU r; for (int i = 0; i < LANES; i++) r.v[i] = i == 63 ? ~0ull : ((2ull << i) - 1); return r; }

// lane i receives lane i-1; lane 0 receives `fill`
inline VI shr1(const VI& x, int fill) { VI r; r.v[0] = fill; for (int i = 1; i < LANES; i++) r.v[i] = x.v[i - 1]; return r; }
// inclusive prefix minimum over lanes 0..i
inline VI prefix_min(const VI& x) { VI r; int m = INF; for (int i = 0; i < LANES; i++) { m = x.v[i] < m ? x.v[i] : m; r.v[i] = m; } return r; }
inline uint64_t ballot(const VB& c) { uint64_t m = 0; for (int i = 0; i < LANES; i++) if (c.v[i]) m |= 1ull << i; return m; }
inline int read_lane(const VI& x, int lane) { return x.v[lane]; }
inline VI write_lane(VI x, int value, int lane) { x.v[lane] = value; return x; }
inline VU make_vu(const VI& lo, const VI& hi) { VU r; for (int i = 0; i < LANES; i++) r.v[i] = ((uint64_t)(uint32_t)hi.v[i] << 32) | (uint32_t)lo.v[i]; return r; }
inline uint64_t read_lane(const VU& x, int lane) { return x.v[lane]; }

// lane i (< count) loads / stores element i; other lanes get `fill` / do nothing
template <typename T> inline VI load_lanes(const T* p, int count, int fill) { VI r; for (int i = 0; i < LANES; i++) r.v[i] = i < count ? (int)p[i] : fill; return r; }
template <typename T> inline void store_lanes(T* p, int count, const VI& x) { for (int i = 0; i < LANES && i < count; i++) p[i] = (T)x.v[i]; }
inline void store_lanes(uint64_t* p, int count, const VU& x) { for (int i = 0; i < LANES && i < count; i++) p[i] = x.v[i]; }
inline VU load_lanes_u64(const uint64_t* p, int count) { VU r; for (int i = 0; i < LANES; i++) r.v[i] = i < count ? p[i] : 0; return r; }
// per-lane indexed load / masked indexed store
template <typename T> inline VI gather(const T* p, const VI& idx) { VI r; for (int i = 0; i < LANES; i++) r.v[i] = (int)p[idx.v[i]]; return r; }
inline VU gather64(const uint64_t* p, const VI& idx) { VU r; for (int i = 0; i < LANES; i++) r.v[i] = p[idx.v[i]]; return r; }
// word `word` of the 16-word record of element idx (64-bit addressing: idx is an unsigned 32-bit element number)
inline VI gather_rec(const uint32_t* p, const VI& idx, int word) { VI r; for (int i = 0; i < LANES; i++) r.v[i] = (int)p[(uint64_t)(uint32_t)idx.v[i] * 16 + (uint32_t)word]; return r; }
template <typename T> inline void scatter(T* p, const VI& idx, const VI& x, const VB& m) { for (int i = 0; i < LANES; i++) if (m.v[i]) p[idx.v[i]] = (T)x.v[i]; }
inline void scatter64(uint64_t* p, const VI& idx, const VU& x, const VB& m) { for (int i = 0; i < LANES; i++) if (m.v[i]) p[idx.v[i]] = x.v[i]; }
// a value every lane holds identically, handed to the scalar unit
inline int wave_uniform(int x) { return x; }
inline void wave_sync() {}
inline void wave_order() {}
inline uint64_t stamp() { return 0; }
// one reservation for the whole wave; every lane sees the old value
inline uint64_t wave_atomic_add(uint64_t* p, uint64_t v) { uint64_t r = *p; *p += v; return r; }
inline uint32_t wave_atomic_add(uint32_t* p, uint32_t v) { uint32_t r = *p; *p += v; return r; }
// claim `bytes` of a pool whose top is *p, only if they fit below `cap`; ~0 when they do not (the top is left alone)
inline uint64_t wave_claim(uint64_t* p, uint64_t bytes, uint64_t cap) { if (*p + bytes > cap) return ~0ull; uint64_t r = *p; *p += bytes; return r; }

}  // namespace gaw
