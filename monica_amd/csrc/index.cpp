// Host side of the reference-genome minimizer index: FASTA(.gz) -> minimizers ->
// sorted (hash, occurrence) arrays -> file.  Replaces what mappy.Aligner(fn_idx_in=<fasta>,
// preset='map-ont', fn_idx_out=<file>) does for monica (aligner.py:45-46) and what
// mappy.Aligner(fn_idx_in=<file>) does on load (aligner.py:59).  The on-disk format is ours.
//
// Index construction is a one-off per database chunk and is not the classification hot
// path; the HBM-resident probe table is derived from these arrays at upload (engine.hip).
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <zlib.h>
#include <sys/stat.h>
#include <stdexcept>

#include "common.h"
#include <parallel/algorithm>

namespace mnc {

static thread_local char g_err[512];

void set_error(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

// ---------------------------------------------------------------- minimizers of one contig
// (w,k)-minimizers with minimap2's tie and boundary behaviour:
// the newest of equal hashes wins the window, every record equal to the window minimum is
// reported, an ambiguous base restarts the k-mer run but not the ring.
static inline uint32_t mix30(uint32_t key, uint32_t mask)
{
	key = (~key + (key << 21)) & mask;
	key ^= key >> 24;
	key = (key + (key << 3) + (key << 8)) & mask;
	key ^= key >> 14;
	key = (key + (key << 2) + (key << 4)) & mask;
	key ^= key >> 28;
	return key;                 // the last round (key += key << 31) vanishes under a <=32-bit mask
}

static inline int base_code(unsigned char c)
{
	switch (c) {
	case 'A': case 'a': return 0;
	case 'C': case 'c': return 1;
	case 'G': case 'g': return 2;
	case 'T': case 't': case 'U': case 'u': return 3;
	default: return 4;
	}
}

namespace {
struct Rec { uint64_t h; uint64_t y; };          // h = hash<<8 | span ; UINT64_MAX = none
constexpr uint64_t NONE = ~0ULL;

class WindowMin {
public:
	WindowMin(int w, std::vector<std::pair<uint64_t, uint64_t>> &out) : w_(w), ring_(w, Rec{NONE, NONE}), out_(out) {}
	void emit(const Rec &r) { out_.emplace_back(r.h >> 8, r.y); }
	void same_as_min(int from, int to) {
		for (int j = from; j < to; ++j)
			if (ring_[j].h == min_.h && ring_[j].y != min_.y) emit(ring_[j]);
	}
	// `run` = number of valid bases seen since the last ambiguous one, `k` = k-mer size
	void push(const Rec &cur, int run, int k) {
		ring_[at_] = cur;
		if (run == w_ + k - 1 && min_.h != NONE) { same_as_min(at_ + 1, w_); same_as_min(0, at_); }
		if (cur.h <= min_.h) {
			if (run >= w_ + k && min_.h != NONE) emit(min_);
			min_ = cur, min_at_ = at_;
		} else if (at_ == min_at_) {
			if (run >= w_ + k - 1 && min_.h != NONE) emit(min_);
			min_.h = NONE;
			for (int j = at_ + 1; j < w_; ++j) if (min_.h >= ring_[j].h) min_ = ring_[j], min_at_ = j;
			for (int j = 0; j <= at_; ++j)     if (min_.h >= ring_[j].h) min_ = ring_[j], min_at_ = j;
			if (run >= w_ + k - 1 && min_.h != NONE) { same_as_min(at_ + 1, w_); same_as_min(0, at_ + 1); }
		}
		if (++at_ == w_) at_ = 0;
	}
	void flush() { if (min_.h != NONE) emit(min_); }
private:
	int w_, at_ = 0, min_at_ = 0;
	std::vector<Rec> ring_;
	Rec min_{NONE, NONE};
	std::vector<std::pair<uint64_t, uint64_t>> &out_;
};
} // namespace

// The minimizers of one contig whose k-mers end in [lo, hi): what the scan of the whole contig reports there.  The
// state of the scan (ring of the last w records, the minimum = the right-most smallest of them, the run of valid bases
// compared with w + k) is a function of the last 2 w + k bases, so a scan started SKETCH_WARM bases early is in the same
// state from `lo` on; a record is reported at the latest when it leaves the window, w bases after it, so the scan runs
// that much past `hi` (what it reports outside [lo, hi) belongs to the neighbours; only the contig's true end flushes).
constexpr int64_t SKETCH_WARM = 64, SKETCH_SEG = 1 << 18;
static void contig_minimizers(const char *s, int64_t len, int64_t lo, int64_t hi, int w, int k, uint32_t rid,
                              std::vector<std::pair<uint64_t, uint64_t>> &out)
{
	const uint32_t mask = (uint32_t)((1ULL << 2 * k) - 1);
	const int top = 2 * (k - 1);
	uint32_t fw = 0, rv = 0;
	int run = 0;
	std::vector<std::pair<uint64_t, uint64_t>> all;
	const bool whole = lo == 0 && hi >= len;
	WindowMin win(w, whole ? out : all);
	const int64_t from = std::max<int64_t>(0, lo - SKETCH_WARM), to = std::min(len, hi + w + 2);
	for (int64_t i = from; i < to; ++i) {
		int c = base_code((unsigned char)s[i]);
		Rec cur{NONE, NONE};
		if (c < 4) {
			fw = (fw << 2 | (uint32_t)c) & mask;
			rv = rv >> 2 | (uint32_t)(3 ^ c) << top;
			if (fw == rv) continue;                         // cannot happen for odd k
			int strand = fw < rv ? 0 : 1;
			++run;
			if (run >= k) {
				cur.h = (uint64_t)mix30(strand ? rv : fw, mask) << 8 | (uint64_t)k;
				cur.y = (uint64_t)rid << 32 | (uint64_t)(uint32_t)i << 1 | (uint64_t)strand;
			}
		} else run = 0;
		win.push(cur, run, k);
	}
	if (to == len) win.flush();
	if (!whole)
		for (auto &pr : all) {
			const int64_t pos = (int64_t)((uint32_t)pr.second >> 1);
			if (pos >= lo && pos < hi) out.push_back(pr);
		}
}

// contig bases -> 4-bit codes.  Words shared by two contigs are written by one thread only:
// a contig's leading and trailing partial words are merged serially afterwards.
void pack_contigs(mnc_index *idx, const char *const *seqs, const int64_t *lens, int n_seq)
{
	uint32_t *S = idx->seq4.data();
#pragma omp parallel for schedule(dynamic, 1)
	for (int i = 0; i < n_seq; ++i) {
		const int64_t o = idx->seq_off[i], e = o + lens[i];
		const int64_t first_full = (o + 7) / 8 * 8, last_full = e / 8 * 8;
		for (int64_t b = first_full; b + 8 <= last_full; b += 8) {
			uint32_t wd = 0;
			for (int t = 0; t < 8; ++t) wd |= (uint32_t)base_code((unsigned char)seqs[i][b - o + t]) << (t * 4);
			S[b >> 3] = wd;
		}
	}
	for (int i = 0; i < n_seq; ++i) {
		const int64_t o = idx->seq_off[i], e = o + lens[i];
		const int64_t first_full = std::min(e, (o + 7) / 8 * 8), last_full = std::max(first_full, e / 8 * 8);
		for (int64_t b = o; b < first_full; ++b) S[b >> 3] |= (uint32_t)base_code((unsigned char)seqs[i][b - o]) << ((b & 7) * 4);
		for (int64_t b = last_full; b < e; ++b) S[b >> 3] |= (uint32_t)base_code((unsigned char)seqs[i][b - o]) << ((b & 7) * 4);
	}
}

// ---------------------------------------------------------------- occurrence cut-off
// value at rank (1 - f) * n_distinct among per-key occurrence counts, plus one (A.3)
int cal_mid_occ(const mnc_index *idx, float f)
{
	size_t n = idx->keys.size();
	if (f <= 0.) return INT32_MAX;
	if (n == 0) return 1;
	std::vector<uint32_t> c(n);
	for (size_t i = 0; i < n; ++i) c[i] = (uint32_t)(idx->key_off[i + 1] - idx->key_off[i]);
	size_t kth = (size_t)(uint32_t)((1. - f) * n);
	if (kth >= n) kth = n - 1;
	std::nth_element(c.begin(), c.begin() + kth, c.end());
	return (int)(c[kth] + 1);
}

// pairs (hash, occurrence word) -> sorted arrays + genome table + mid_occ
int index_finalize(mnc_index *idx, std::vector<std::pair<uint64_t, uint64_t>> &pairs)
{
	__gnu_parallel::sort(pairs.begin(), pairs.end());      // (hash, occurrence word): a total order
	idx->keys.clear(), idx->key_off.clear(), idx->pos.resize(pairs.size());
	for (size_t i = 0; i < pairs.size(); ++i) {
		if (i == 0 || pairs[i].first != pairs[i - 1].first) {
			idx->keys.push_back((uint32_t)pairs[i].first);
			idx->key_off.push_back(i);
		}
		idx->pos[i] = pairs[i].second;
	}
	idx->key_off.push_back(pairs.size());
	index_genome_table(idx);
	idx->mid_occ = cal_mid_occ(idx, idx->par.mid_occ_frac);
	return MNC_OK;
}

// genomes = distinct contig names in order of first appearance (database.py:59-64 gives
// every record of one genome the same "tax_unit:accession" id)
void index_genome_table(mnc_index *idx)
{
	std::map<std::string, int> seen;
	idx->contig_genome.assign(idx->contig_name.size(), 0);
	idx->genome_name.clear(), idx->genome_len.clear();
	idx->total_len = 0;
	for (size_t i = 0; i < idx->contig_name.size(); ++i) {
		auto it = seen.find(idx->contig_name[i]);
		int g;
		if (it == seen.end()) {
			g = (int)idx->genome_name.size();
			seen.emplace(idx->contig_name[i], g);
			idx->genome_name.push_back(idx->contig_name[i]);
			idx->genome_len.push_back(0);
		} else g = it->second;
		idx->contig_genome[i] = g;
		idx->genome_len[g] += idx->contig_len[i];
		idx->total_len += idx->contig_len[i];
	}
}

static int check_kw(int k, int w)
{
	if (k != 15 || w != 10) {
		set_error("only k=15, w=10 (minimap2 'map-ont', the setting monica uses) is implemented; got k=%d w=%d", k, w);
		return MNC_ERR_UNSUPPORTED;
	}
	return MNC_OK;
}

} // namespace mnc

using namespace mnc;

// ================================================================ C-ABI: errors / misc

extern "C" const char *mnc_strerror(int code)
{
	switch (code) {
	case MNC_OK: return "ok";
	case MNC_ERR_ARG: return "invalid argument";
	case MNC_ERR_IO: return "i/o error";
	case MNC_ERR_FORMAT: return "Damaged or empty index";
	case MNC_ERR_NOMEM: return "out of memory";
	case MNC_ERR_HIP: return "HIP runtime error";
	case MNC_ERR_NODEVICE: return "no usable gfx950 device";
	case MNC_ERR_UNSUPPORTED: return "unsupported parameter";
	case MNC_ERR_RANGE: return "buffer too small";
	default: return "unknown error";
	}
}

extern "C" const char *mnc_last_error(void) { return g_err; }
extern "C" const char *mnc_version(void) { return "monica_amd 0.1 (gfx950)"; }

// ================================================================ C-ABI: index

extern "C" int mnc_index_build_mem(int n_seq, const char *const *names, const char *const *seqs,
                                   const int64_t *lens, int k, int w, mnc_index **out)
{
	if (!out || n_seq < 0 || (n_seq > 0 && (!names || !seqs || !lens))) return MNC_ERR_ARG;
	*out = nullptr;
	if (int e = check_kw(k, w)) return e;
	mnc_index *idx = new (std::nothrow) mnc_index;
	if (!idx) return MNC_ERR_NOMEM;
	idx->k = k, idx->w = w;
	std::vector<std::pair<uint64_t, uint64_t>> pairs;
	try {
		for (int i = 0; i < n_seq; ++i) {
			if (lens[i] < 0 || lens[i] > 0x7fffffffLL) { delete idx; set_error("contig %d too long", i); return MNC_ERR_UNSUPPORTED; }
			idx->contig_name.emplace_back(names[i]);
			idx->contig_len.push_back(lens[i]);
		}
		// pieces of contigs are independent (contig_minimizers): sketch them on all host threads
		struct Piece { int contig; int64_t lo, hi; };
		std::vector<Piece> pieces;
		for (int i = 0; i < n_seq; ++i)
			for (int64_t lo = 0; lo == 0 || lo < lens[i]; lo += SKETCH_SEG) pieces.push_back({ i, lo, lo + SKETCH_SEG >= lens[i] ? lens[i] : lo + SKETCH_SEG });
		std::vector<std::vector<std::pair<uint64_t, uint64_t>>> per(pieces.size());
		bool oom = false;
		idx->seq_off.assign((size_t)n_seq + 1, 0);
		for (int i = 0; i < n_seq; ++i) idx->seq_off[i + 1] = idx->seq_off[i] + lens[i];
		idx->seq4.assign((size_t)(idx->seq_off[n_seq] + 7) / 8 + 1, 0u);
#pragma omp parallel for schedule(dynamic, 1)
		for (size_t j = 0; j < pieces.size(); ++j) {
			const Piece &pc = pieces[j];
			try {
				per[j].reserve((size_t)((pc.hi - pc.lo) / 5 + 16));
				contig_minimizers(seqs[pc.contig], lens[pc.contig], pc.lo, pc.hi, w, k, (uint32_t)pc.contig, per[j]);
			} catch (const std::bad_alloc &) { oom = true; }
		}
		pack_contigs(idx, seqs, lens, n_seq);
		if (oom) { delete idx; return MNC_ERR_NOMEM; }
		size_t total = 0;
		for (auto &v : per) total += v.size();
		pairs.reserve(total);
		for (auto &v : per) { pairs.insert(pairs.end(), v.begin(), v.end()); std::vector<std::pair<uint64_t, uint64_t>>().swap(v); }
		index_finalize(idx, pairs);
	} catch (const std::bad_alloc &) { delete idx; return MNC_ERR_NOMEM; }
	*out = idx;
	return MNC_OK;
}

static int index_build_from_fasta(const char *fasta_path, const char *out_path, int k, int w, int device, mnc_index **out)
{
	if (!fasta_path || !out) return MNC_ERR_ARG;
	*out = nullptr;
	if (int e = check_kw(k, w)) return e;
	gzFile fp = gzopen(fasta_path, "rb");
	if (!fp) { set_error("cannot open %s", fasta_path); return MNC_ERR_IO; }
	gzbuffer(fp, 1 << 20);
	std::vector<std::string> names, seqs;
	std::vector<char> line(1 << 16);
	try {
		while (gzgets(fp, line.data(), (int)line.size())) {
			size_t L = strlen(line.data());
			bool eol = L && line[L - 1] == '\n';
			while (L && (line[L - 1] == '\n' || line[L - 1] == '\r')) --L;
			if (line[0] == '>' && (seqs.empty() || true)) {
				// header: name = text up to the first whitespace (Appendix A.8)
				size_t e = 1;
				while (e < L && line[e] != ' ' && line[e] != '\t') ++e;
				names.emplace_back(line.data() + 1, e - 1);
				seqs.emplace_back();
				while (!eol && gzgets(fp, line.data(), (int)line.size())) {   // swallow an over-long header
					size_t l2 = strlen(line.data());
					eol = l2 && line[l2 - 1] == '\n';
				}
			} else if (!names.empty()) {
				seqs.back().append(line.data(), L);
			}
		}
	} catch (const std::bad_alloc &) { gzclose(fp); return MNC_ERR_NOMEM; }
	gzclose(fp);
	if (names.empty()) { set_error("%s holds no FASTA record", fasta_path); return MNC_ERR_FORMAT; }
	std::vector<const char*> np(names.size()), sp(names.size());
	std::vector<int64_t> lp(names.size());
	for (size_t i = 0; i < names.size(); ++i) np[i] = names[i].c_str(), sp[i] = seqs[i].data(), lp[i] = (int64_t)seqs[i].size();
	mnc_index *idx = nullptr;
	int e = device < 0 ? mnc_index_build_mem((int)names.size(), np.data(), sp.data(), lp.data(), k, w, &idx)
	                   : mnc_index_build_mem_device((int)names.size(), np.data(), sp.data(), lp.data(), k, w, device, &idx);
	if (e) return e;
	if (idx->keys.empty()) { delete idx; set_error("no minimizer in %s", fasta_path); return MNC_ERR_FORMAT; }
	if (out_path) {
		e = mnc_index_save(idx, out_path);
		if (e) { delete idx; return e; }
	}
	*out = idx;
	return MNC_OK;
}

extern "C" int mnc_index_build(const char *fasta_path, const char *out_path, int k, int w, mnc_index **out)
{
	return index_build_from_fasta(fasta_path, out_path, k, w, -1, out);
}

extern "C" int mnc_index_build_device(const char *fasta_path, const char *out_path, int k, int w, int device, mnc_index **out)
{
	if (device < 0) return MNC_ERR_ARG;
	return index_build_from_fasta(fasta_path, out_path, k, w, device, out);
}

namespace {
const char MAGIC[8] = { 'M', 'N', 'C', 'I', 'D', 'X', '2', 0 };   // 2: contig bases follow the occurrence words
struct FileHeader {
	char magic[8];
	int32_t k, w, n_contigs, mid_occ;
	int64_t n_keys, n_occ, names_bytes;
};
template <class T> bool put(FILE *f, const T *p, size_t n) { return n == 0 || fwrite(p, sizeof(T), n, f) == n; }
template <class T> bool get(FILE *f, T *p, size_t n) { return n == 0 || fread(p, sizeof(T), n, f) == n; }
}

extern "C" int mnc_index_save(const mnc_index *idx, const char *path)
{
	if (!idx || !path) return MNC_ERR_ARG;
	FILE *f = fopen(path, "wb");
	if (!f) { set_error("cannot create %s", path); return MNC_ERR_IO; }
	std::string names;
	for (auto &s : idx->contig_name) names.append(s), names.push_back('\0');
	FileHeader h;
	memcpy(h.magic, MAGIC, 8);
	h.k = idx->k, h.w = idx->w, h.n_contigs = (int32_t)idx->contig_name.size(), h.mid_occ = idx->mid_occ;
	h.n_keys = (int64_t)idx->keys.size(), h.n_occ = (int64_t)idx->pos.size(), h.names_bytes = (int64_t)names.size();
	bool ok = put(f, &h, 1) && put(f, names.data(), names.size()) &&
	          put(f, idx->contig_len.data(), idx->contig_len.size()) &&
	          put(f, idx->keys.data(), idx->keys.size()) &&
	          put(f, idx->key_off.data(), idx->key_off.size()) &&
	          put(f, idx->pos.data(), idx->pos.size()) &&
	          put(f, idx->seq4.data(), idx->seq4.size());
	ok = (fclose(f) == 0) && ok;
	if (!ok) { set_error("short write to %s", path); return MNC_ERR_IO; }
	return MNC_OK;
}


// ---------------------------------------------------------------- minimap2's index file
// The file mappy writes at aligner.py:45-46 (`fn_idx_out`) and reads at aligner.py:59: "MMI\2",
// u32 {w, k, b, n_seq, flag}; per sequence u8 name length, name, u32 length; per bucket (2^b of them,
// bucket = hash & (2^b - 1)) i32 n + n occurrence words of its repeated minimizers, u32 size + size
// pairs {key = hash >> b << 1 | singleton, value = the occurrence word | offset << 32 | count}; then
// the bases, 4 bits each, 8 per u32 (SURVEY.md A.3 -- stated there from the published format of
// minimap2 2.17; no file written by minimap2 itself was at hand to check it against).
namespace {
constexpr int MMI_NO_SEQ = 2, MMI_HPC = 1;
struct MmiHead { uint32_t w, k, b, n_seq, flag; };
}

extern "C" int mnc_index_save_mmi(const mnc_index *idx, const char *path)
{
	if (!idx || !path) return MNC_ERR_ARG;
	for (auto &s : idx->contig_name)
		if (s.size() > 255) { set_error("contig name longer than 255 bytes: the .mmi format cannot hold it"); return MNC_ERR_UNSUPPORTED; }
	FILE *f = fopen(path, "wb");
	if (!f) { set_error("cannot create %s", path); return MNC_ERR_IO; }
	const int b = 14;
	const MmiHead h = { (uint32_t)idx->w, (uint32_t)idx->k, (uint32_t)b, (uint32_t)idx->contig_name.size(), 0u };
	bool ok = fwrite("MMI\2", 1, 4, f) == 4 && put(f, &h, 1);
	for (size_t i = 0; ok && i < idx->contig_name.size(); ++i) {
		const uint8_t l = (uint8_t)idx->contig_name[i].size();
		const uint32_t len = (uint32_t)idx->contig_len[i];
		ok = put(f, &l, 1) && put(f, idx->contig_name[i].data(), l) && put(f, &len, 1);
	}
	// keys by bucket (ascending hash inside one)
	const size_t nb = (size_t)1 << b, nk = idx->keys.size();
	std::vector<uint32_t> start(nb + 1, 0), order(nk);
	for (size_t i = 0; i < nk; ++i) ++start[(idx->keys[i] & (nb - 1)) + 1];
	for (size_t i = 0; i < nb; ++i) start[i + 1] += start[i];
	{
		std::vector<uint32_t> cur(start.begin(), start.end() - 1);
		for (size_t i = 0; i < nk; ++i) order[cur[idx->keys[i] & (nb - 1)]++] = (uint32_t)i;
	}
	std::vector<uint64_t> p, kv;
	for (size_t bi = 0; ok && bi < nb; ++bi) {
		p.clear(), kv.clear();
		for (uint32_t o = start[bi]; o < start[bi + 1]; ++o) {
			const size_t i = order[o];
			const uint64_t cnt = idx->key_off[i + 1] - idx->key_off[i], hi = (uint64_t)(idx->keys[i] >> b) << 1;
			if (cnt == 1) kv.push_back(hi | 1), kv.push_back(idx->pos[idx->key_off[i]]);
			else {
				kv.push_back(hi), kv.push_back((uint64_t)p.size() << 32 | cnt);
				p.insert(p.end(), idx->pos.begin() + (ptrdiff_t)idx->key_off[i], idx->pos.begin() + (ptrdiff_t)idx->key_off[i + 1]);
			}
		}
		const int32_t n = (int32_t)p.size();
		const uint32_t size = (uint32_t)(kv.size() / 2);
		ok = put(f, &n, 1) && put(f, p.data(), p.size()) && put(f, &size, 1) && put(f, kv.data(), kv.size());
	}
	ok = ok && put(f, idx->seq4.data(), (size_t)(idx->total_len + 7) / 8);
	ok = (fclose(f) == 0) && ok;
	if (!ok) { set_error("short write to %s", path); return MNC_ERR_IO; }
	return MNC_OK;
}

static int load_mmi(FILE *f, const char *path, int64_t file_size, mnc_index **out)
{
	auto bad = [&](const char *what) { set_error("%s is truncated or damaged (%s)", path, what); return MNC_ERR_FORMAT; };
	MmiHead h;
	if (!get(f, &h, 1)) return bad("header");
	if (h.b < 1 || h.b > 28 || h.n_seq == 0 || (int64_t)h.n_seq * 5 > file_size) return bad("header");
	if (h.flag & MMI_HPC) { set_error("%s: homopolymer-compressed indexes are not supported", path); return MNC_ERR_UNSUPPORTED; }
	if (h.flag & MMI_NO_SEQ) { set_error("%s holds no sequences: base-level alignment needs them", path); return MNC_ERR_UNSUPPORTED; }
	if (int e = check_kw((int)h.k, (int)h.w)) return e;
	mnc_index *idx = new (std::nothrow) mnc_index;
	if (!idx) return MNC_ERR_NOMEM;
	int rc = MNC_OK;
	try {
		idx->k = (int)h.k, idx->w = (int)h.w;
		int64_t total = 0;
		for (uint32_t i = 0; i < h.n_seq && !rc; ++i) {
			uint8_t l;
			char name[256];
			uint32_t len;
			if (!get(f, &l, 1) || !get(f, name, l) || !get(f, &len, 1) || len > 0x7fffffffu) { rc = bad("sequence table"); break; }
			idx->contig_name.emplace_back(name, l);
			idx->contig_len.push_back((int64_t)len);
			total += len;
		}
		std::vector<std::pair<uint64_t, uint64_t>> pairs;
		std::vector<uint64_t> p, kv;
		const int64_t rest_min = (total + 7) / 8 * 4;
		if (file_size > rest_min) pairs.reserve((size_t)((file_size - rest_min) / 8));      // every occurrence takes at least 8 bytes of the file
		for (uint64_t bi = 0; bi < (1ull << h.b) && !rc; ++bi) {
			int32_t n;
			uint32_t size;
			if (!get(f, &n, 1) || n < 0 || (int64_t)n * 8 > file_size) { rc = bad("bucket"); break; }
			p.resize((size_t)n);
			if (!get(f, p.data(), p.size()) || !get(f, &size, 1) || (int64_t)size * 16 > file_size) { rc = bad("bucket"); break; }
			kv.resize((size_t)size * 2);
			if (!get(f, kv.data(), kv.size())) { rc = bad("bucket"); break; }
			for (uint32_t j = 0; j < size; ++j) {
				const uint64_t key = kv[2 * j], val = kv[2 * j + 1];
				const uint64_t hash = (key >> 1) << h.b | bi;
				if (hash >= (1ull << 30)) { rc = bad("minimizer beyond 2k bits"); break; }
				if (key & 1) pairs.emplace_back(hash, val);
				else {
					const uint64_t off = val >> 32, cnt = (uint32_t)val;
					if (cnt < 2 || off + cnt > (uint64_t)n) { rc = bad("occurrence list"); break; }
					for (uint64_t t = 0; t < cnt; ++t) pairs.emplace_back(hash, p[off + t]);
				}
			}
		}
		if (!rc) {
			const int64_t here = (int64_t)ftello(f);
			if (here < 0 || file_size - here < rest_min) rc = bad("sequences");
		}
		if (!rc) {
			idx->seq_off.assign((size_t)h.n_seq + 1, 0);
			for (uint32_t i = 0; i < h.n_seq; ++i) idx->seq_off[i + 1] = idx->seq_off[i] + idx->contig_len[i];
			idx->seq4.assign((size_t)(total + 7) / 8 + 1, 0u);
			if (!get(f, idx->seq4.data(), (size_t)(total + 7) / 8)) rc = bad("sequences");
		}
		for (size_t i = 0; !rc && i < pairs.size(); ++i) {
			const uint64_t rid = pairs[i].second >> 32, ps = (uint32_t)pairs[i].second >> 1;
			if (rid >= h.n_seq || (int64_t)ps >= idx->contig_len[rid]) rc = bad("occurrence word");
		}
		if (!rc && pairs.empty()) rc = bad("no minimizer");
		if (!rc) {
			index_finalize(idx, pairs);                 // sorts, groups and derives mid_occ as mappy does on loading
			for (size_t k = 0; !rc && k < idx->keys.size(); ++k)    // an occurrence listed twice under one minimizer: not minimap2's
				for (uint64_t o = idx->key_off[k] + 1; o < idx->key_off[k + 1]; ++o)
					if (idx->pos[o] == idx->pos[o - 1]) { rc = bad("duplicate occurrence"); break; }
		}
	} catch (const std::bad_alloc &) { rc = MNC_ERR_NOMEM; }
	if (rc) { delete idx; return rc; }
	*out = idx;
	return MNC_OK;
}

extern "C" int mnc_index_load(const char *path, mnc_index **out)
{
	if (!path || !out) return MNC_ERR_ARG;
	*out = nullptr;
	FILE *f = fopen(path, "rb");
	if (!f) { set_error("cannot open %s", path); return MNC_ERR_IO; }
	{   // minimap2's own format (what an installation that ran the reference holds)?
		char m4[4];
		if (fread(m4, 1, 4, f) == 4 && memcmp(m4, "MMI\2", 4) == 0) {
			struct stat sb;
			if (fstat(fileno(f), &sb) != 0) { fclose(f); set_error("cannot stat %s", path); return MNC_ERR_IO; }
			const int rc = load_mmi(f, path, (int64_t)sb.st_size, out);
			fclose(f);
			return rc;
		}
		rewind(f);
	}
	FileHeader h;
	if (!get(f, &h, 1) || memcmp(h.magic, MAGIC, 8) != 0 || h.n_contigs <= 0 || h.n_keys <= 0 ||
	    h.n_occ < h.n_keys || h.names_bytes <= 0) {
		fclose(f);
		set_error("%s is not a monica_amd index (or is empty)", path);
		return MNC_ERR_FORMAT;
	}
	// the header must account for the file, byte for byte, before anything is allocated from it
	struct stat sb;
	if (fstat(fileno(f), &sb) != 0) { fclose(f); set_error("cannot stat %s", path); return MNC_ERR_IO; }
	const unsigned __int128 fixed = (unsigned __int128)sizeof(FileHeader) + (unsigned __int128)h.names_bytes +
	                                (unsigned __int128)h.n_contigs * 8 + (unsigned __int128)h.n_keys * 4 +
	                                ((unsigned __int128)h.n_keys + 1) * 8 + (unsigned __int128)h.n_occ * 8;
	if (fixed > (unsigned __int128)sb.st_size || ((unsigned __int128)sb.st_size - fixed) % 4 != 0) {
		fclose(f);
		set_error("%s is truncated or damaged (header does not match the file size)", path);
		return MNC_ERR_FORMAT;
	}
	const size_t seq_words = (size_t)(((unsigned __int128)sb.st_size - fixed) / 4);
	mnc_index *idx = new (std::nothrow) mnc_index;
	if (!idx) { fclose(f); return MNC_ERR_NOMEM; }
	int rc = MNC_OK;
	try {
		std::string names((size_t)h.names_bytes, '\0');
		idx->k = h.k, idx->w = h.w;
		idx->contig_len.resize((size_t)h.n_contigs);
		idx->keys.resize((size_t)h.n_keys);
		idx->key_off.resize((size_t)h.n_keys + 1);
		idx->pos.resize((size_t)h.n_occ);
		idx->seq4.resize(seq_words);
		bool ok = get(f, &names[0], names.size()) && get(f, idx->contig_len.data(), idx->contig_len.size()) &&
		          get(f, idx->keys.data(), idx->keys.size()) && get(f, idx->key_off.data(), idx->key_off.size()) &&
		          get(f, idx->pos.data(), idx->pos.size()) && get(f, idx->seq4.data(), idx->seq4.size());
		if (ok) {
			const char *p = names.data(), *e = p + names.size();
			while (p < e && (int)idx->contig_name.size() < h.n_contigs) {
				const void *z = memchr(p, 0, (size_t)(e - p));
				if (!z) break;
				idx->contig_name.emplace_back(p);
				p = (const char*)z + 1;
			}
			ok = (int)idx->contig_name.size() == h.n_contigs && p == e;
		}
		// the arrays index each other and, later, device memory: check every invariant here
		// rather than fault in a kernel
		int64_t total = 0;
		for (size_t i = 0; ok && i < idx->contig_len.size(); ++i) {
			ok = idx->contig_len[i] >= 0 && idx->contig_len[i] <= 0x7fffffffLL;
			total += idx->contig_len[i];
		}
		ok = ok && seq_words == (size_t)(total + 7) / 8 + 1;
		ok = ok && idx->key_off[0] == 0 && idx->key_off.back() == (uint64_t)h.n_occ;
		for (size_t i = 0; ok && i < idx->keys.size(); ++i)
			ok = idx->key_off[i] < idx->key_off[i + 1] && idx->keys[i] < (1u << 30) && (i == 0 || idx->keys[i - 1] < idx->keys[i]);
		for (size_t i = 0; ok && i < idx->pos.size(); ++i) {
			const uint64_t rid = idx->pos[i] >> 32, ps = (uint32_t)idx->pos[i] >> 1;
			ok = rid < (uint64_t)h.n_contigs && (int64_t)ps < idx->contig_len[rid];
		}
		if (!ok) { set_error("%s is truncated or damaged", path); rc = MNC_ERR_FORMAT; }
		else {
			idx->seq_off.assign((size_t)h.n_contigs + 1, 0);
			for (int i = 0; i < h.n_contigs; ++i) idx->seq_off[i + 1] = idx->seq_off[i] + idx->contig_len[i];
			// rebuild genome table; keep the stored cut-off
			std::vector<std::pair<uint64_t, uint64_t>> none;
			std::vector<uint32_t> keys; std::vector<uint64_t> off, pos;
			keys.swap(idx->keys), off.swap(idx->key_off), pos.swap(idx->pos);
			index_finalize(idx, none);
			keys.swap(idx->keys), off.swap(idx->key_off), pos.swap(idx->pos);
			if (h.mid_occ < 1) { set_error("%s is truncated or damaged", path); rc = MNC_ERR_FORMAT; }
			idx->mid_occ = h.mid_occ;
			if (!rc) rc = check_kw(idx->k, idx->w);
		}
	} catch (const std::bad_alloc &) { rc = MNC_ERR_NOMEM; }
	catch (const std::exception &ex) { set_error("%s is truncated or damaged (%s)", path, ex.what()); rc = MNC_ERR_FORMAT; }
	fclose(f);
	if (rc) { delete idx; return rc; }
	*out = idx;
	return MNC_OK;
}

extern "C" void mnc_index_free(mnc_index *idx)
{
	if (!idx) return;
	index_release_device(idx);
	delete idx;
}

extern "C" int mnc_index_info(const mnc_index *idx, mnc_index_info_t *info)
{
	if (!idx || !info) return MNC_ERR_ARG;
	memset(info, 0, sizeof(*info));
	info->k = idx->k, info->w = idx->w;
	info->n_contigs = (int32_t)idx->contig_name.size();
	info->n_genomes = (int32_t)idx->genome_name.size();
	info->mid_occ = idx->mid_occ;
	info->n_keys = (int64_t)idx->keys.size();
	info->n_occ = (int64_t)idx->pos.size();
	info->total_len = idx->total_len;
	for (auto &d : idx->dev) { info->device_bytes += d.bytes; info->table_slots = ((int64_t)1 << d.pb_bits) << d.region_bits; }
	return MNC_OK;
}

// HBM the index holds on ONE device (mnc_index_info sums over every device it is resident on)
extern "C" int mnc_index_device_bytes(const mnc_index *idx, int device, int64_t *bytes)
{
	if (!idx || !bytes) return MNC_ERR_ARG;
	*bytes = 0;
	std::lock_guard<std::mutex> lk(const_cast<mnc_index*>(idx)->dev_mutex);
	for (auto &d : idx->dev) if (d.device == device) *bytes += d.bytes;
	return MNC_OK;
}

extern "C" const char *mnc_index_contig_name(const mnc_index *idx, int rid)
{
	return idx && rid >= 0 && rid < (int)idx->contig_name.size() ? idx->contig_name[rid].c_str() : nullptr;
}
extern "C" int64_t mnc_index_contig_len(const mnc_index *idx, int rid)
{
	return idx && rid >= 0 && rid < (int)idx->contig_len.size() ? idx->contig_len[rid] : -1;
}
extern "C" int mnc_index_contig_genome(const mnc_index *idx, int rid)
{
	return idx && rid >= 0 && rid < (int)idx->contig_genome.size() ? idx->contig_genome[rid] : -1;
}
extern "C" const char *mnc_index_genome_name(const mnc_index *idx, int gid)
{
	return idx && gid >= 0 && gid < (int)idx->genome_name.size() ? idx->genome_name[gid].c_str() : nullptr;
}
extern "C" int64_t mnc_index_genome_len(const mnc_index *idx, int gid)
{
	return idx && gid >= 0 && gid < (int)idx->genome_len.size() ? idx->genome_len[gid] : -1;
}

extern "C" int mnc_index_dump(const mnc_index *idx, uint64_t *hash, uint64_t *y, int64_t cap, int64_t *n)
{
	if (!idx || !n) return MNC_ERR_ARG;
	*n = (int64_t)idx->pos.size();
	if (cap < *n || !hash || !y) return MNC_ERR_RANGE;
	for (size_t i = 0; i < idx->keys.size(); ++i)
		for (uint64_t j = idx->key_off[i]; j < idx->key_off[i + 1]; ++j) hash[j] = idx->keys[i], y[j] = idx->pos[j];
	return MNC_OK;
}

extern "C" int mnc_index_set_mid_occ(mnc_index *idx, int mid_occ)
{
	if (!idx || mid_occ < 1) return MNC_ERR_ARG;
	idx->mid_occ = mid_occ;
	return MNC_OK;
}

// ================================================================ C-ABI: host-side monica layer

// aligner.py:328-339 in exact integer arithmetic.  float(NM)/mlen in binary64 orders two
// hits exactly like the rationals NM/mlen (distinct rationals with denominators < 2^26
// never round to the same double), so cross-multiplication in int64 is equivalent; the
// reference's "distance == 0 at the last update" rule reduces to "the minimum is attained
// more than once".
extern "C" int mnc_best_hit(const mnc_hit_t *hits, int n, int *best_index)
{
	if (!hits || !best_index || n <= 0) return MNC_ERR_ARG;
	int best = 0, ties = 1;
	for (int i = 1; i < n; ++i) {
		int64_t l = (int64_t)hits[i].nm * hits[best].mlen, r = (int64_t)hits[best].nm * hits[i].mlen;
		if (l < r) best = i, ties = 1;
		else if (l == r) best = i, ++ties;
	}
	*best_index = (n == 1 || ties == 1) ? best : -1;
	return MNC_OK;
}

// aligner.py:247-263
extern "C" int mnc_counts(const mnc_index *idx, const int32_t *assign, const mnc_hit_t *best,
                          const int64_t *offsets, uint32_t n_reads, int mode, int64_t *counts)
{
	if (!idx || !assign || !counts || mode < 1 || mode > 3) return MNC_ERR_ARG;
	if ((mode == 2 && !offsets) || (mode == 3 && !best)) return MNC_ERR_ARG;
	for (uint32_t r = 0; r < n_reads; ++r) {
		if (assign[r] < 0) continue;
		if (assign[r] >= (int)idx->contig_genome.size()) return MNC_ERR_ARG;
		int g = idx->contig_genome[assign[r]];
		counts[g] += mode == 1 ? 1 : mode == 2 ? offsets[r + 1] - offsets[r] : best[r].mlen;
	}
	return MNC_OK;
}
