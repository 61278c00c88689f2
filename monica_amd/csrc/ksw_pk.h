// ksw2's cell update (ksw_extd2_sse: two-piece affine gaps in the difference recurrence of Suzuki & Kasahara) on PACKED
// 16-bit pairs: two cells per 32-bit register, the recurrence on VOP3P (v_pk_add_u16 / v_pk_sub_u16 / v_pk_max_i16 /
// v_pk_min_i16).  This is what index.map() runs per region in the reference (monica/genomes/aligner.py:193, 215 -> mappy
// 2.17 -> mm_align1 -> ksw_extd2_sse); the kernel that uses it (csrc/k_align.hip: ksw_wp) keeps ksw2's own array layout,
// 16-lane rounding and stale-cell semantics, so it answers the calls whose band clips the matrix.
//
// A 16-bit lane is  [ value : int8 in bits 15..8 | tag : bits 7..0 ].
//   * The value byte is ksw2's int8 itself: a 16-bit add / sub wraps its high byte exactly as _mm_add_epi8 /
//     _mm_sub_epi8 wrap a byte, provided the tag bytes never carry or borrow -- they do not: every add has one operand
//     with a zero tag, every sub a subtrahend with a zero tag.  A signed 16-bit max / min orders by the signed value
//     first and by the tag second.
//   * The tag byte does the work of ksw2's comparison chains.  Bits 7..4 hold a RANK per candidate of
//     H = max(s, a, b, a2, b2): ksw2 lets the first of equal candidates win when gaps are left-aligned (a later one must
//     be strictly greater) and the last when KSW_EZ_RIGHT is set -- the ranks fall resp. rise along the chain, so the
//     maximum itself picks ksw2's winner and carries its name.  Bits 3..0 hold one "extended" bit per gap state
//     (x: 1, y: 2, x2: 4, y2: 8): the state's new value is max(a - tmp, 0) - qe in ksw2, here max(a - (z + e), -(q + e))
//     against a constant whose tag differs from the state's in that one bit, on the side that gives ksw2's `a > 0`
//     (left) or `a >= 0` (right) on a tie.  (The two forms agree when no intermediate leaves int8: kpk::params_fit.)
//   * The direction byte of a cell is the XOR of the five tag bytes (two v_bitop3): an invertible code of ksw2's byte,
//     kpk::decode turns it back where the walk reads it.
//
// The file compiles for the host as well (plain C++ emulation of the packed operations): tests/test_ksw_packed.py runs the
// array-level form of the kernel against the oracle's literal simulation on the CPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define KPK_FN __host__ __device__ __forceinline__
#else
#define KPK_FN static inline
#endif

namespace kpk {

#if defined(__HIP_DEVICE_COMPILE__)
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
KPK_FN uint32_t add(uint32_t x, uint32_t y) { return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, x) + __builtin_bit_cast(u16x2, y))); }
KPK_FN uint32_t sub(uint32_t x, uint32_t y) { return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, x) - __builtin_bit_cast(u16x2, y))); }
KPK_FN uint32_t maxs(uint32_t x, uint32_t y) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, x), __builtin_bit_cast(s16x2, y))); }
KPK_FN uint32_t mins(uint32_t x, uint32_t y) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(s16x2, x), __builtin_bit_cast(s16x2, y))); }
KPK_FN uint32_t minu(uint32_t x, uint32_t y) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, x), __builtin_bit_cast(u16x2, y))); }
KPK_FN uint32_t madu(uint32_t x, uint32_t y, uint32_t z) { return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, x) * __builtin_bit_cast(u16x2, y) + __builtin_bit_cast(u16x2, z))); }
KPK_FN uint32_t shift16(uint32_t hi, uint32_t lo) { return __builtin_amdgcn_alignbit(hi, lo, 16); }        // {lo.hi16, hi.lo16}: every cell's neighbour below
KPK_FN uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return (uint32_t)__builtin_amdgcn_bitop3_b32((int)a, (int)b, (int)c, 0x96); }
#else
KPK_FN uint32_t pk2(uint32_t lo, uint32_t hi) { return (lo & 0xffffu) | hi << 16; }
KPK_FN uint32_t add(uint32_t x, uint32_t y) { return pk2(x + y, (x >> 16) + (y >> 16)); }
KPK_FN uint32_t sub(uint32_t x, uint32_t y) { return pk2(x - y, (x >> 16) - (y >> 16)); }
KPK_FN int16_t s16(uint32_t v) { return (int16_t)(uint16_t)v; }
KPK_FN uint32_t maxs(uint32_t x, uint32_t y) { return pk2(s16(x) > s16(y) ? x : y, s16(x >> 16) > s16(y >> 16) ? x >> 16 : y >> 16); }
KPK_FN uint32_t mins(uint32_t x, uint32_t y) { return pk2(s16(x) < s16(y) ? x : y, s16(x >> 16) < s16(y >> 16) ? x >> 16 : y >> 16); }
KPK_FN uint32_t minu(uint32_t x, uint32_t y) { return pk2((x & 0xffffu) < (y & 0xffffu) ? x : y, (x >> 16) < (y >> 16) ? x >> 16 : y >> 16); }
KPK_FN uint32_t madu(uint32_t x, uint32_t y, uint32_t z) { return pk2((x & 0xffffu) * (y & 0xffffu) + z, (x >> 16) * (y >> 16) + (z >> 16)); }
KPK_FN uint32_t shift16(uint32_t hi, uint32_t lo) { return lo >> 16 | hi << 16; }
KPK_FN uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return a ^ b ^ c; }
#endif
KPK_FN uint32_t rep(int v) { return ((uint32_t)v & 0xffffu) * 0x10001u; }
KPK_FN uint32_t lane(int value, int tag) { return rep((int)(((uint32_t)value & 0xffu) << 8 | (uint32_t)tag)); }
KPK_FN uint32_t bitsel(uint32_t mask, uint32_t x, uint32_t y) { return (x & mask) | (y & ~mask); }
KPK_FN int value_of(uint32_t v, int half) { return (int)(int8_t)(uint8_t)(half ? v >> 24 : v >> 8); }

// ---- tags.  rank << 4 | extended bits
template <bool RIGHT> struct Tags {
	static constexpr int RS = RIGHT ? 1 : 5, RA = RIGHT ? 2 : 4, RB = 3, RA2 = RIGHT ? 4 : 2, RB2 = RIGHT ? 5 : 1;
	// the state as it is kept (and as it enters the next cell's sums) / the constant it is compared with
	static constexpr int X = RA << 4 | (RIGHT ? 1 : 0), KX = RA << 4 | (RIGHT ? 0 : 1);
	static constexpr int Y = RB << 4 | (RIGHT ? 2 : 0), KY = RB << 4 | (RIGHT ? 0 : 2);
	static constexpr int X2 = RA2 << 4 | (RIGHT ? 4 : 0), KX2 = RA2 << 4 | (RIGHT ? 0 : 4);
	static constexpr int Y2 = RB2 << 4 | (RIGHT ? 8 : 0), KY2 = RB2 << 4 | (RIGHT ? 0 : 8);
	static constexpr int S = RS << 4;
};

// every intermediate of the cell stays inside int8 (then max(a - tmp, 0) - qe == max(a - (z + e), -(q + e)), and no
// value ever wraps): u, v in [-(q + e), mch + q + e], x, y in [-(q + e), -e], x2, y2 in [-(q2 + e2), -e2], z in [mis, mch]
KPK_FN bool params_fit(int q, int e, int q2, int e2, int mch, int mis /* < 0 */, int sc_n /* < 0 */)
{
	if (q < 0 || e < 0 || q2 < 0 || e2 < 0 || mch < 0 || mis > 0 || sc_n > 0) return false;
	const int g = (q + e > q2 + e2 ? q + e : q2 + e2), lo = -mis > -sc_n ? -mis : -sc_n;
	return 2 * g + mch + lo + (q > q2 ? q : q2) + (e > e2 ? e : e2) < 120 && q + e <= q2 + e2;
}

struct Consts {
	uint32_t s_match, s_delta;         // the score of equal bases (tagged) / what unequal ones add to it
	uint32_t mch8, e8, e28;            // clean: tag 0
	uint32_t kx, ky, kx2, ky2;         // -(q + e), -(q2 + e2) with the constants' tags
	uint32_t ix, iy, ix2, iy2, iuv, is; // what ksw2 fills its arrays with before the first anti-diagonal
};
template <bool RIGHT> KPK_FN Consts make_consts(int q, int e, int q2, int e2, int mch, int mis)
{
	typedef Tags<RIGHT> T;
	Consts K;
	K.s_match = lane(mch, T::S), K.s_delta = lane(mis - mch, 0);
	K.mch8 = lane(mch, 0), K.e8 = lane(e, 0), K.e28 = lane(e2, 0);
	K.kx = lane(-q - e, T::KX), K.ky = lane(-q - e, T::KY), K.kx2 = lane(-q2 - e2, T::KX2), K.ky2 = lane(-q2 - e2, T::KY2);
	K.ix = lane(-q - e, T::X), K.iy = lane(-q - e, T::Y), K.ix2 = lane(-q2 - e2, T::X2), K.iy2 = lane(-q2 - e2, T::Y2);
	K.iuv = lane(-q - e, 0), K.is = lane(0, T::S);
	return K;
}

// the scores of two cells from their base codes (0..3 in the low bits of each half; ambiguous codes are not for this form)
KPK_FN uint32_t scores(const Consts &K, uint32_t tb, uint32_t qb) { return madu(minu(tb ^ qb, 0x00010001u), K.s_delta, K.s_match); }

// Two cells of an anti-diagonal.  xb, vb, x2b: x, v, x2 of the cells BELOW them (t - 1), old; u, v, x, y, x2, y2: the
// cells' own, old on entry and new on return; s: their scores.  Returns the direction code in the low byte of each half.
template <bool RIGHT>
KPK_FN uint32_t cell_pair(const Consts &K, uint32_t xb, uint32_t vb, uint32_t x2b, uint32_t s,
                          uint32_t &u, uint32_t &v, uint32_t &x, uint32_t &y, uint32_t &x2, uint32_t &y2)
{
	const uint32_t a = add(xb, vb), b = add(y, u), a2 = add(x2b, vb), b2 = add(y2, u);
	const uint32_t zt = maxs(maxs(maxs(maxs(s, a), b), a2), b2);
	const uint32_t z = mins(zt & 0xff00ff00u, K.mch8);
	const uint32_t un = sub(z, vb), vn = sub(z, u);
	const uint32_t tz = add(z, K.e8), tz2 = add(z, K.e28);
	const uint32_t xd = maxs(sub(a, tz), K.kx), yd = maxs(sub(b, tz), K.ky), x2d = maxs(sub(a2, tz2), K.kx2), y2d = maxs(sub(b2, tz2), K.ky2);
	u = un, v = vn;
	if (RIGHT) x = xd | 0x00010001u, y = yd | 0x00020002u, x2 = x2d | 0x00040004u, y2 = y2d | 0x00080008u;
	else x = xd & 0xfffefffeu, y = yd & 0xfffdfffdu, x2 = x2d & 0xfffbfffbu, y2 = y2d & 0xfff7fff7u;
	return xor3(xor3(xd, yd, x2d), y2d, zt);
}

// the direction code of one cell -> ksw2's byte: bits 0-2 which of {0 diagonal, 1 x, 2 y, 3 x2, 4 y2} gave H, bits 3-6
// "the x / y / x2 / y2 state of the next cell extends this one" (ksw_backtrack reads them)
template <bool RIGHT> KPK_FN uint32_t decode(uint32_t code)
{
	typedef Tags<RIGHT> T;
	const uint32_t rank = (code >> 4 & 15u) ^ (uint32_t)(T::RA ^ T::RB ^ T::RA2 ^ T::RB2);
	const uint32_t which = RIGHT ? rank - 1u : 5u - rank;
	uint32_t lo = code & 15u;
	if (RIGHT) { if (which > 0) lo ^= 1u << (which - 1u); }
	else lo = ~lo & 15u;
	return which | lo << 3;
}

} // namespace kpk
