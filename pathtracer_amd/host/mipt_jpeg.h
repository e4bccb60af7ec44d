// mipt_jpeg.h — JPEG decoding for the OBJ / scene loaders: baseline and progressive Huffman JPEG
// (ITU T.81), 8-bit, 1 component (grey) or 3 components (YCbCr, or RGB when the component ids / the Adobe marker say so).
//
// The reference decodes textures with stb_image (utils.cpp:105), and a texel feeds the radiance bit for bit, so the
// numeric choices of that decoder are the specification here:
//   * coefficients are dequantised into 16-bit integers (products wrap);
//   * inverse DCT = the IJG "islow" integer transform with 12-bit constants, columns first with 2 extra bits kept
//     (>> 10 after +512), rows with +65536 + (128 << 17) and >> 17, then clamped to 0..255;
//   * chroma upsampling = the triangle filter: (3 near + far + 2) >> 2 for a factor 2 in one direction,
//     (3 t_near + t_far + 8) >> 4 on the vertically filtered rows for 2 x 2; nearest neighbour for other factors;
//     a row of the image takes the chroma row above / below according to its position in the pair;
//   * YCbCr -> RGB in 20-bit fixed point with the coefficients 1.40200, 0.71414, 0.34414, 1.77200 rounded to 12 bits,
//     the Cb term of green truncated to its upper 16 bits.
// Arithmetic-coded, lossless, 12-bit and 4-component (CMYK) files are refused.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace mipt_jpeg {

struct Huff {
	uint8_t bits[17] = {0};
	uint8_t vals[256] = {0};
	int mincode[18] = {0}, maxcode[18], valptr[17] = {0};
	bool present = false;                   // set by build(); a scan that names a table no DHT defined is refused (scan_header)
	Huff() { for (int& m : maxcode) m = -1; }   // no code length matches until build() ran
	void build() {
		int code = 0, k = 0;
		for (int l = 1; l <= 16; l++) {
			valptr[l] = k; mincode[l] = code;
			code += bits[l]; k += bits[l];
			maxcode[l] = bits[l] ? code - 1 : -1;
			code <<= 1;
		}
		present = true;
	}
};

struct Comp {
	int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0, dc_pred = 0;
	int x = 0, y = 0, w2 = 0, h2 = 0;       // size in samples, size padded to whole MCUs
	std::vector<uint8_t> data;              // w2 x h2 samples
	std::vector<int16_t> coeff;             // progressive: 64 per block, (w2/8) x (h2/8) blocks
};

struct Decoder {
	const uint8_t* p; const uint8_t* end;
	std::string err;
	uint16_t dequant[4][64];
	Huff hdc[4], hac[4];
	Comp c[4];
	int ncomp = 0, W = 0, H = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
	bool progressive = false, jfif = false, rgb_ids = false;
	int adobe_transform = -1, restart_interval = 0;
	// scan state
	uint32_t bitbuf = 0; int nbits = 0; bool hit_marker = false; int marker = 0;
	int spec_start = 0, spec_end = 63, succ_high = 0, succ_low = 0, eob_run = 0, todo = 0;
	int scan_n = 0, order[4];

	bool fail(const char* m) { if (err.empty()) err = m; return false; }
	int get8() { return p < end ? *p++ : 0; }
	int get16() { int a = get8(); return (a << 8) | get8(); }

	void fill() {                                     // keep at least 25 bits; a marker inside the data stops the feed with zeros
		while (nbits <= 24) {
			int b = hit_marker ? 0 : get8();
			if (b == 0xff && !hit_marker) {
				int m = get8();
				while (m == 0xff) m = get8();
				if (m != 0) { marker = m; hit_marker = true; b = 0; }
			}
			bitbuf |= (uint32_t)b << (24 - nbits);
			nbits += 8;
		}
	}
	int getbits(int n) { if (n == 0) return 0; if (nbits < n) fill(); int v = (int)(bitbuf >> (32 - n)); bitbuf <<= n; nbits -= n; return v; }
	int getbit() { return getbits(1); }
	int decode(const Huff& h) {
		if (nbits < 16) fill();
		int code = 0;
		for (int l = 1; l <= 16; l++) {
			code = (int)(bitbuf >> (32 - l));
			if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) {
				bitbuf <<= l; nbits -= l;
				return h.vals[h.valptr[l] + code - h.mincode[l]];
			}
		}
		return -1;
	}
	int extend(int s) {                               // RECEIVE + EXTEND (F.2.2.1)
		if (s == 0) return 0;
		int v = getbits(s);
		return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
	}
	void reset_scan() {
		bitbuf = 0; nbits = 0; hit_marker = false; marker = 0; eob_run = 0;
		for (int i = 0; i < 4; i++) c[i].dc_pred = 0;
		todo = restart_interval ? restart_interval : 0x7fffffff;
	}

	static const uint8_t* zigzag() {
		static const uint8_t z[64 + 15] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
		                                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
		                                   63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};
		return z;
	}

	// ---- inverse DCT (IJG jidctint with 12-bit constants) --------------------------------------------------------
	static int f2f(double x) { return (int)(x * 4096 + 0.5); }
	static uint8_t clamp8(int x) { return x < 0 ? 0 : (x > 255 ? 255 : (uint8_t)x); }
	struct Row { int x0, x1, x2, x3, t0, t1, t2, t3; };
	static Row idct1d(int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7) {
		Row r;
		int p2 = s2, p3 = s6;
		int p1 = (p2 + p3) * f2f(0.5411961f);
		int t2 = p1 + p3 * f2f(-1.847759065f);
		int t3 = p1 + p2 * f2f(0.765366865f);
		p2 = s0; p3 = s4;
		int t0 = (p2 + p3) * 4096, t1 = (p2 - p3) * 4096;
		r.x0 = t0 + t3; r.x3 = t0 - t3; r.x1 = t1 + t2; r.x2 = t1 - t2;
		t0 = s7; t1 = s5; t2 = s3; t3 = s1;
		p3 = t0 + t2; int p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;
		int p5 = (p3 + p4) * f2f(1.175875602f);
		t0 = t0 * f2f(0.298631336f); t1 = t1 * f2f(2.053119869f); t2 = t2 * f2f(3.072711026f); t3 = t3 * f2f(1.501321110f);
		p1 = p5 + p1 * f2f(-0.899976223f); p2 = p5 + p2 * f2f(-2.562915447f);
		p3 = p3 * f2f(-1.961570560f); p4 = p4 * f2f(-0.390180644f);
		r.t3 = t3 + p1 + p4; r.t2 = t2 + p2 + p3; r.t1 = t1 + p2 + p4; r.t0 = t0 + p1 + p3;
		return r;
	}
	static void idct(uint8_t* out, int stride, const int16_t* d) {
		int val[64];
		for (int i = 0; i < 8; i++) {
			Row r = idct1d(d[i], d[8 + i], d[16 + i], d[24 + i], d[32 + i], d[40 + i], d[48 + i], d[56 + i]);
			r.x0 += 512; r.x1 += 512; r.x2 += 512; r.x3 += 512;
			val[i] = (r.x0 + r.t3) >> 10; val[56 + i] = (r.x0 - r.t3) >> 10;
			val[8 + i] = (r.x1 + r.t2) >> 10; val[48 + i] = (r.x1 - r.t2) >> 10;
			val[16 + i] = (r.x2 + r.t1) >> 10; val[40 + i] = (r.x2 - r.t1) >> 10;
			val[24 + i] = (r.x3 + r.t0) >> 10; val[32 + i] = (r.x3 - r.t0) >> 10;
		}
		for (int i = 0; i < 8; i++) {
			const int* v = val + 8 * i;
			uint8_t* o = out + (size_t)stride * i;
			Row r = idct1d(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
			const int bias = 65536 + (128 << 17);
			r.x0 += bias; r.x1 += bias; r.x2 += bias; r.x3 += bias;
			o[0] = clamp8((r.x0 + r.t3) >> 17); o[7] = clamp8((r.x0 - r.t3) >> 17);
			o[1] = clamp8((r.x1 + r.t2) >> 17); o[6] = clamp8((r.x1 - r.t2) >> 17);
			o[2] = clamp8((r.x2 + r.t1) >> 17); o[5] = clamp8((r.x2 - r.t1) >> 17);
			o[3] = clamp8((r.x3 + r.t0) >> 17); o[4] = clamp8((r.x3 - r.t0) >> 17);
		}
	}

	// ---- entropy-coded blocks ------------------------------------------------------------------------------------
	bool block_baseline(int16_t* data, Comp& cp) {
		const uint16_t* dq = dequant[cp.tq];
		int t = decode(hdc[cp.hd]);
		if (t < 0 || t > 15) return fail("bad huffman code");     // a DC category above 15 (crafted DHT) would shift by more than the bit buffer holds
		memset(data, 0, 64 * sizeof(int16_t));
		int dc = cp.dc_pred + (t ? extend(t) : 0);
		cp.dc_pred = dc;
		data[0] = (int16_t)(dc * dq[0]);
		int k = 1;
		do {
			int rs = decode(hac[cp.ha]);
			if (rs < 0) return fail("bad huffman code");
			int s = rs & 15, r = rs >> 4;
			if (s == 0) { if (rs != 0xf0) break; k += 16; }
			else { k += r; int z = zigzag()[k++]; data[z] = (int16_t)(extend(s) * dq[z]); }
		} while (k < 64);
		return true;
	}
	bool block_prog_dc(int16_t* data, Comp& cp) {
		if (spec_end != 0) return fail("can't merge dc and ac");
		if (succ_high == 0) {
			memset(data, 0, 64 * sizeof(int16_t));
			int t = decode(hdc[cp.hd]);
			if (t < 0 || t > 15) return fail("bad huffman code");
			int dc = cp.dc_pred + (t ? extend(t) : 0);
			cp.dc_pred = dc;
			data[0] = (int16_t)(dc << succ_low);
		} else if (getbit()) data[0] += (int16_t)(1 << succ_low);
		return true;
	}
	bool block_prog_ac(int16_t* data, Comp& cp) {
		if (spec_start == 0) return fail("can't merge dc and ac");
		const Huff& h = hac[cp.ha];
		if (succ_high == 0) {
			const int shift = succ_low;
			if (eob_run) { --eob_run; return true; }
			int k = spec_start;
			do {
				int rs = decode(h);
				if (rs < 0) return fail("bad huffman code");
				int s = rs & 15, r = rs >> 4;
				if (s == 0) {
					if (r < 15) { eob_run = 1 << r; if (r) eob_run += getbits(r); --eob_run; break; }
					k += 16;
				} else { k += r; int z = zigzag()[k++]; data[z] = (int16_t)(extend(s) << shift); }
			} while (k <= spec_end);
		} else {
			const int16_t bit = (int16_t)(1 << succ_low);
			if (eob_run) {
				--eob_run;
				for (int k = spec_start; k <= spec_end; k++) {
					int16_t* q = &data[zigzag()[k]];
					if (*q != 0 && getbit() && (*q & bit) == 0) { if (*q > 0) *q += bit; else *q -= bit; }
				}
			} else {
				int k = spec_start;
				do {
					int rs = decode(h);
					if (rs < 0) return fail("bad huffman code");
					int s = rs & 15, r = rs >> 4;
					if (s == 0) {
						if (r < 15) { eob_run = (1 << r) - 1; if (r) eob_run += getbits(r); r = 64; }   // force the end of the block
					} else {
						if (s != 1) return fail("bad huffman code");
						s = getbit() ? bit : -bit;
					}
					while (k <= spec_end) {
						int16_t* q = &data[zigzag()[k++]];
						if (*q != 0) {
							if (getbit() && (*q & bit) == 0) { if (*q > 0) *q += bit; else *q -= bit; }
						} else {
							if (r == 0) { *q = (int16_t)s; break; }
							--r;
						}
					}
				} while (k <= spec_end);
			}
		}
		return true;
	}

	bool restart_if_due() {                             // after every MCU (or block of a single-component scan)
		if (--todo <= 0) {
			if (nbits < 24) fill();
			if (!(marker >= 0xd0 && marker <= 0xd7)) return false;   // no restart marker where one is due: the scan ends here
			reset_scan();
		}
		return true;
	}

	bool scan() {
		reset_scan();
		int16_t tmp[64];
		if (scan_n == 1) {                              // single component: its own blocks in raster order
			Comp& cp = c[order[0]];
			const int w = (cp.x + 7) >> 3, h = (cp.y + 7) >> 3;
			for (int j = 0; j < h; j++) for (int i = 0; i < w; i++) {
				if (!progressive) {
					if (!block_baseline(tmp, cp)) return false;
					idct(&cp.data[(size_t)cp.w2 * j * 8 + i * 8], cp.w2, tmp);
				} else {
					int16_t* d = &cp.coeff[64 * ((size_t)i + (size_t)j * (cp.w2 / 8))];
					if (!(spec_start == 0 ? block_prog_dc(d, cp) : block_prog_ac(d, cp))) return false;
				}
				if (!restart_if_due()) return true;
			}
			return true;
		}
		for (int j = 0; j < mcuy; j++) for (int i = 0; i < mcux; i++) {   // interleaved MCUs
			for (int k = 0; k < scan_n; k++) {
				Comp& cp = c[order[k]];
				for (int y = 0; y < cp.v; y++) for (int x = 0; x < cp.h; x++) {
					const int x2 = (i * cp.h + x), y2 = (j * cp.v + y);
					if (!progressive) {
						if (!block_baseline(tmp, cp)) return false;
						idct(&cp.data[(size_t)cp.w2 * y2 * 8 + x2 * 8], cp.w2, tmp);
					} else {
						if (!block_prog_dc(&cp.coeff[64 * ((size_t)x2 + (size_t)y2 * (cp.w2 / 8))], cp)) return false;
					}
				}
			}
			if (!restart_if_due()) return true;
		}
		return true;
	}

	void finish_progressive() {                         // dequantise and transform the collected coefficients
		for (int n = 0; n < ncomp; n++) {
			Comp& cp = c[n];
			const int w = (cp.x + 7) >> 3, h = (cp.y + 7) >> 3;
			for (int j = 0; j < h; j++) for (int i = 0; i < w; i++) {
				int16_t* d = &cp.coeff[64 * ((size_t)i + (size_t)j * (cp.w2 / 8))];
				for (int k = 0; k < 64; k++) d[k] = (int16_t)(d[k] * dequant[cp.tq][k]);
				idct(&cp.data[(size_t)cp.w2 * j * 8 + i * 8], cp.w2, d);
			}
		}
	}

	// ---- markers ------------------------------------------------------------------------------------------------
	bool frame_header(int m) {
		int L = get16();
		if (get8() != 8) return fail("only 8-bit JPEG is decoded");
		H = get16(); W = get16();
		if (W <= 0 || H <= 0) return fail("bad image size");
		if ((size_t)W * (size_t)H > ((size_t)1 << 28)) return fail("image too large");
		ncomp = get8();
		if (ncomp != 1 && ncomp != 3) return fail("only grey and 3-component JPEG is decoded (no CMYK)");
		if (L != 8 + 3 * ncomp) return fail("bad SOF length");
		rgb_ids = true;
		static const char rgb[3] = {'R', 'G', 'B'};
		for (int i = 0; i < ncomp; i++) {
			c[i].id = get8();
			if (ncomp != 3 || c[i].id != rgb[i]) rgb_ids = false;
			int q = get8(); c[i].h = q >> 4; c[i].v = q & 15;
			if (c[i].h < 1 || c[i].h > 4 || c[i].v < 1 || c[i].v > 4) return fail("bad sampling factor");
			c[i].tq = get8(); if (c[i].tq > 3) return fail("bad quantisation table index");
		}
		hmax = vmax = 1;
		for (int i = 0; i < ncomp; i++) { if (c[i].h > hmax) hmax = c[i].h; if (c[i].v > vmax) vmax = c[i].v; }
		// the upsampling walks whole multiples (hs = hmax / h): factors that do not divide the maximum (h = 3 under hmax = 4)
		// would read W samples from rows that hold fewer
		for (int i = 0; i < ncomp; i++) if (hmax % c[i].h != 0 || vmax % c[i].v != 0) return fail("sampling factors that do not divide the largest one");
		mcux = (W + hmax * 8 - 1) / (hmax * 8); mcuy = (H + vmax * 8 - 1) / (vmax * 8);
		for (int i = 0; i < ncomp; i++) {
			c[i].x = (W * c[i].h + hmax - 1) / hmax; c[i].y = (H * c[i].v + vmax - 1) / vmax;
			c[i].w2 = mcux * c[i].h * 8; c[i].h2 = mcuy * c[i].v * 8;
			c[i].data.assign((size_t)c[i].w2 * c[i].h2, 0);
			if (m == 0xc2) c[i].coeff.assign((size_t)c[i].w2 * c[i].h2, 0);
		}
		progressive = (m == 0xc2);
		return true;
	}
	bool scan_header() {
		get16();
		scan_n = get8();
		if (scan_n < 1 || scan_n > ncomp) return fail("bad SOS component count");
		for (int i = 0; i < scan_n; i++) {
			int id = get8(), q = get8(), which = -1;
			for (int k = 0; k < ncomp; k++) if (c[k].id == id) which = k;
			if (which < 0) return fail("bad SOS component");
			c[which].hd = q >> 4; c[which].ha = q & 15;
			if (c[which].hd > 3 || c[which].ha > 3) return fail("bad huffman table index");
			order[i] = which;
		}
		spec_start = get8(); spec_end = get8();
		int a = get8(); succ_high = a >> 4; succ_low = a & 15;
		if (progressive) { if (spec_start > 63 || spec_end > 63 || spec_start > spec_end || succ_high > 13 || succ_low > 13) return fail("bad SOS"); }
		else { if (spec_start != 0 || succ_high != 0 || succ_low != 0) return fail("bad SOS"); spec_end = 63; }
		for (int i = 0; i < scan_n; i++) {                    // every table the scan will decode with must have come in a DHT
			const Comp& cp = c[order[i]];
			const bool need_dc = !progressive || (spec_start == 0 && succ_high == 0), need_ac = !progressive || spec_start > 0;
			if ((need_dc && !hdc[cp.hd].present) || (need_ac && !hac[cp.ha].present)) return fail("scan uses a huffman table that was never defined");
		}
		return true;
	}
	bool other_marker(int m) {
		int L = get16() - 2;
		if (L < 0) return fail("bad marker length");
		const uint8_t* seg_end = p + L;
		if (seg_end > end) return fail("truncated file");
		if (m == 0xdb) {                                // DQT
			while (p < seg_end) {
				int q = get8(), sixteen = q >> 4, t = q & 15;
				if (sixteen > 1 || t > 3) return fail("bad DQT");
				for (int i = 0; i < 64; i++) dequant[t][zigzag()[i]] = (uint16_t)(sixteen ? get16() : get8());
			}
		} else if (m == 0xc4) {                         // DHT
			while (p < seg_end) {
				int q = get8(), tc = q >> 4, th = q & 15, n = 0;
				if (tc > 1 || th > 3) return fail("bad DHT");
				Huff& h = tc ? hac[th] : hdc[th];
				for (int i = 1; i <= 16; i++) { h.bits[i] = (uint8_t)get8(); n += h.bits[i]; }
				if (n > 256) return fail("bad DHT");
				for (int i = 0; i < n; i++) h.vals[i] = (uint8_t)get8();
				h.build();
			}
		} else if (m == 0xdd) restart_interval = get16();
		else if (m == 0xe0 && L >= 5 && !memcmp(p, "JFIF\0", 5)) jfif = true;
		else if (m == 0xee && L >= 12 && !memcmp(p, "Adobe\0", 6)) adobe_transform = p[11];
		p = seg_end;
		return true;
	}

	bool decode_image(const uint8_t* data, size_t n) {
		p = data; end = data + n;
		if (get8() != 0xff || get8() != 0xd8) return fail("not a JPEG file");
		bool have_frame = false;
		int m = next_marker();
		for (;;) {
			if (m < 0) return fail("truncated file");
			if (m == 0xd9) break;
			if (m == 0xc0 || m == 0xc1 || m == 0xc2) { if (have_frame) return fail("two frames"); if (!frame_header(m)) return false; have_frame = true; m = next_marker(); }
			else if (m == 0xc3 || (m >= 0xc5 && m <= 0xcf && m != 0xc8 && m != 0xcc)) return fail("lossless / hierarchical / arithmetic-coded JPEG is not decoded");
			else if (m == 0xda) {
				if (!have_frame) return fail("scan before frame");
				if (!scan_header() || !scan()) return false;
				if (hit_marker) { m = marker; hit_marker = false; }       // the marker the bit reader ran into
				else m = next_marker();
				while (m >= 0xd0 && m <= 0xd7) m = next_marker();         // stray restart markers
			} else { if (!other_marker(m)) return false; m = next_marker(); }
		}
		if (!have_frame) return fail("no frame");
		if (progressive) finish_progressive();
		return true;
	}
	int next_marker() {
		while (p < end) {
			int b = get8();
			if (b != 0xff) continue;
			int m = get8();
			while (m == 0xff) m = get8();
			if (m != 0) return m;
		}
		return -1;
	}

	// ---- upsampling and colour conversion ------------------------------------------------------------------------
	void to_rgb(std::vector<unsigned char>& rgb) {
		rgb.resize((size_t)W * H * 3);
		struct Res { int hs, vs, ystep, ypos, w_lores; const uint8_t *line0, *line1; std::vector<uint8_t> buf; } r[3];
		for (int k = 0; k < ncomp; k++) {
			r[k].hs = hmax / c[k].h; r[k].vs = vmax / c[k].v; r[k].ystep = r[k].vs >> 1; r[k].ypos = 0;
			r[k].w_lores = (W + r[k].hs - 1) / r[k].hs;
			r[k].line0 = r[k].line1 = c[k].data.data();
			r[k].buf.resize((size_t)W + 8);
		}
		const bool is_rgb = ncomp == 3 && (rgb_ids || (adobe_transform == 0 && !jfif));
		for (int j = 0; j < H; j++) {
			const uint8_t* co[3] = {nullptr, nullptr, nullptr};
			for (int k = 0; k < ncomp; k++) {
				Res& q = r[k];
				const bool y_bot = q.ystep >= (q.vs >> 1);
				const uint8_t* nr = y_bot ? q.line1 : q.line0; const uint8_t* fr = y_bot ? q.line0 : q.line1;
				uint8_t* o = q.buf.data();
				const int w = q.w_lores;
				if (q.hs == 1 && q.vs == 1) co[k] = nr;
				else if (q.hs == 1 && q.vs == 2) { for (int i = 0; i < w; i++) o[i] = (uint8_t)((3 * nr[i] + fr[i] + 2) >> 2); co[k] = o; }
				else if (q.hs == 2 && q.vs == 1) {
					if (w == 1) o[0] = o[1] = nr[0];
					else {
						o[0] = nr[0]; o[1] = (uint8_t)((nr[0] * 3 + nr[1] + 2) >> 2);
						int i;
						for (i = 1; i < w - 1; i++) { const int n = 3 * nr[i] + 2; o[i * 2] = (uint8_t)((n + nr[i - 1]) >> 2); o[i * 2 + 1] = (uint8_t)((n + nr[i + 1]) >> 2); }
						o[i * 2] = (uint8_t)((nr[w - 2] * 3 + nr[w - 1] + 2) >> 2); o[i * 2 + 1] = nr[w - 1];
					}
					co[k] = o;
				} else if (q.hs == 2 && q.vs == 2) {
					if (w == 1) o[0] = o[1] = (uint8_t)((3 * nr[0] + fr[0] + 2) >> 2);
					else {
						int t1 = 3 * nr[0] + fr[0];
						o[0] = (uint8_t)((t1 + 2) >> 2);
						for (int i = 1; i < w; i++) { const int t0 = t1; t1 = 3 * nr[i] + fr[i]; o[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4); o[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4); }
						o[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
					}
					co[k] = o;
				} else {
					q.buf.resize((size_t)w * q.hs + 8); o = q.buf.data();
					for (int i = 0; i < w; i++) for (int a = 0; a < q.hs; a++) o[i * q.hs + a] = nr[i];
					co[k] = o;
				}
				if (++q.ystep >= q.vs) { q.ystep = 0; q.line0 = q.line1; if (++q.ypos < c[k].y) q.line1 += c[k].w2; }
			}
			unsigned char* out = &rgb[(size_t)j * W * 3];
			if (ncomp == 1) { for (int i = 0; i < W; i++) { out[0] = out[1] = out[2] = co[0][i]; out += 3; } }
			else if (is_rgb) { for (int i = 0; i < W; i++) { out[0] = co[0][i]; out[1] = co[1][i]; out[2] = co[2][i]; out += 3; } }
			else {
				auto fx = [](float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; };
				for (int i = 0; i < W; i++) {
					const int yf = (co[0][i] << 20) + (1 << 19);
					const int cr = co[2][i] - 128, cb = co[1][i] - 128;
					int rr = yf + cr * fx(1.40200f);
					int gg = yf + (cr * -fx(0.71414f)) + (int)((unsigned)(cb * -fx(0.34414f)) & 0xffff0000u);
					int bb = yf + cb * fx(1.77200f);
					rr >>= 20; gg >>= 20; bb >>= 20;
					out[0] = clamp8(rr); out[1] = clamp8(gg); out[2] = clamp8(bb);
					out += 3;
				}
			}
		}
	}
};

inline bool decode(const unsigned char* data, size_t n, std::vector<unsigned char>& rgb, int& W, int& H, std::string& why) {
	Decoder d;
	memset(d.dequant, 0, sizeof d.dequant);
	if (!d.decode_image(data, n)) { why = d.err.empty() ? "corrupt JPEG" : d.err; return false; }
	W = d.W; H = d.H;
	d.to_rgb(rgb);
	return true;
}

}   // namespace mipt_jpeg
