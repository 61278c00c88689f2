// Host side of the aligner loop that is not arithmetic: FASTQ batches in, routed FASTQ
// records out, and the per-read hits carried across index parts.
//
// Replaces, behind the C-ABI of include/monica_amd.h:
//   SeqIO.parse(sample, 'fastq') / str(seq_record.seq)     monica/genomes/aligner.py:191-193, 212-215
//   SeqIO.write(seq_record, <mapped|unmapped|ambiguous|focus>, 'fastq')   aligner.py:232-243, 265
//   sample_hits / <sample>_hits.pkl                        aligner.py:184-188, 196-203, 218-223, 267-273
//
// The reference builds one Biopython SeqRecord per read; here a batch is three flat arrays
// (bases, qualities, offsets) plus a title arena, so the Python above it touches only
// batch-level arrays.  The record grammar and the error messages follow Biopython's
// FastqGeneralIterator / FastqPhredIterator / FastqPhredWriter (the library the reference
// calls; not part of this image).
#include "common.h"

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/stat.h>
#include <algorithm>
#include <unistd.h>
#include <unordered_map>

using namespace mnc;

namespace {

constexpr size_t IO_CHUNK = 8u << 20;

struct LineReader {
	int fd = -1;
	std::vector<char> buf;
	size_t lo = 0, hi = 0;
	bool eof = false;
	int err = 0;                // errno of a failed read(): reported as MNC_ERR_IO, never taken for the end of the file

	// next line without its terminator; false at end of file.  `raw_len` = bytes consumed
	bool next(const char *&p, size_t &len)
	{
		for (;;) {
			const char *nl = hi > lo ? (const char*)memchr(buf.data() + lo, '\n', hi - lo) : nullptr;
			if (nl) {
				p = buf.data() + lo;
				len = (size_t)(nl - p);
				lo += len + 1;
				return true;
			}
			if (eof) {
				if (hi == lo) return false;
				p = buf.data() + lo, len = hi - lo, lo = hi;
				return true;
			}
			// refill: keep the partial line, append a chunk
			if (lo > 0) { memmove(buf.data(), buf.data() + lo, hi - lo); hi -= lo, lo = 0; }
			if (buf.size() < hi + IO_CHUNK) buf.resize(hi + IO_CHUNK);
			ssize_t n;
			do n = read(fd, buf.data() + hi, IO_CHUNK); while (n < 0 && errno == EINTR);
			if (n < 0) { err = errno ? errno : EIO; eof = true; return false; }
			if (n == 0) eof = true;
			hi += (size_t)n;
		}
	}
};

inline size_t rstrip_len(const char *p, size_t len)
{
	while (len > 0) {
		const unsigned char c = (unsigned char)p[len - 1];
		if (c == ' ' || (c >= 9 && c <= 13)) --len; else break;
	}
	return len;
}

struct HostBuf {               // growable byte buffer, optionally page-locked
	uint8_t *p = nullptr;
	size_t cap = 0;
	bool pinned = false;
	bool ensure(size_t need, size_t keep)
	{
		if (need <= cap) return true;
		size_t nc = cap ? cap : (1u << 20);
		while (nc < need) nc += nc / 2 + (1u << 20);
		uint8_t *np_ = pinned ? (uint8_t*)mnc_host_alloc(nc) : (uint8_t*)malloc(nc);
		if (!np_) return false;
		if (keep) memcpy(np_, p, keep);
		release();
		p = np_, cap = nc;
		return true;
	}
	void release()
	{
		if (p) { if (pinned) mnc_host_free(p); else free(p); }
		p = nullptr, cap = 0;
	}
};

} // namespace

struct mnc_fastq {
	LineReader in;
	std::string path;
	int64_t file_size = 0;
	bool have_pending = false, started = false, done = false;
	std::string pending;             // the '@' line read ahead by the previous record
	// current batch
	uint32_t n = 0;
	HostBuf bases, quals;
	std::vector<int64_t> offsets;
	std::string titles;
	std::vector<uint64_t> title_off;  // n + 1
	std::vector<uint32_t> id_len, id_off;   // seq_record.id = first word of the title
};

static int fq_io_fail(const mnc_fastq *fq)
{
	set_error("read of %s failed: %s", fq->path.c_str(), strerror(fq->in.err));
	return MNC_ERR_IO;
}

static int fq_fail(const char *msg)
{
	set_error("%s", msg);
	return MNC_ERR_FORMAT;
}

extern "C" int mnc_fastq_open(const char *path, mnc_fastq **out)
{
	if (!path || !out) return MNC_ERR_ARG;
	const int fd = open(path, O_RDONLY);
	if (fd < 0) { set_error("cannot open %s: %s", path, strerror(errno)); return MNC_ERR_IO; }
	mnc_fastq *fq = new mnc_fastq();
	fq->in.fd = fd;
	fq->path = path;
	{
		struct stat sb;
		fq->file_size = fstat(fd, &sb) == 0 ? (int64_t)sb.st_size : 0;
	}
	fq->bases.pinned = true;
	fq->offsets.push_back(0);
	fq->title_off.push_back(0);
	*out = fq;
	return MNC_OK;
}

extern "C" void mnc_fastq_close(mnc_fastq *fq)
{
	if (!fq) return;
	if (fq->in.fd >= 0) close(fq->in.fd);
	fq->bases.release();
	fq->quals.release();
	delete fq;
}

extern "C" int mnc_fastq_next(mnc_fastq *fq, uint32_t max_reads, uint64_t max_bases, uint32_t *n_reads)
{
	if (!fq || !n_reads) return MNC_ERR_ARG;
	fq->n = 0;
	fq->offsets.assign(1, 0);
	fq->titles.clear();
	fq->title_off.assign(1, 0);
	fq->id_len.clear();
	fq->id_off.clear();
	*n_reads = 0;
	if (fq->done || max_reads == 0) return MNC_OK;
	const char *p;
	size_t len;
	if (!fq->started) {
		fq->started = true;
		if (!fq->in.next(p, len)) {
			if (fq->in.err) return fq_io_fail(fq);
			fq->done = true;                                                   // empty file
			return MNC_OK;
		}
		fq->pending.assign(p, len);
		fq->have_pending = true;
	}
	int64_t nb = 0;
	if (!fq->bases.p && fq->file_size > 0) {
		// one allocation for the whole run: at most half of what is left of the file is sequence
		const uint64_t want = std::min<uint64_t>(max_bases, (uint64_t)fq->file_size / 2) + (1u << 20);
		if (!fq->bases.ensure((size_t)want, 0) || !fq->quals.ensure((size_t)want, 0)) { set_error("out of host memory"); return MNC_ERR_NOMEM; }
	}
	while (fq->have_pending) {
		// ---- title
		const std::string &tl = fq->pending;
		if (tl.empty() || tl[0] != '@') return fq_fail("Records in Fastq files should start with '@' character");
		const size_t t_len = rstrip_len(tl.data() + 1, tl.size() - 1);
		const size_t t_off = fq->titles.size();
		fq->titles.append(tl.data() + 1, t_len);
		fq->have_pending = false;
		// ---- sequence lines up to the '+' line
		const int64_t s_off = nb;
		bool first = true;
		for (;;) {
			if (!fq->in.next(p, len)) {
				if (fq->in.err) return fq_io_fail(fq);
				if (first) { first = false; continue; }         // readline() == '' once, then the '+' search fails
				return fq_fail("End of file without quality information.");
			}
			if (!first && len > 0 && p[0] == '+') {
				const size_t c_len = rstrip_len(p + 1, len - 1);
				if (c_len > 0 && (c_len != t_len || memcmp(p + 1, fq->titles.data() + t_off, t_len) != 0))
					return fq_fail("Sequence and quality captions differ.");
				break;
			}
			first = false;
			const size_t l = rstrip_len(p, len);
			if (!fq->bases.ensure((size_t)nb + l + 64, (size_t)nb)) { set_error("out of host memory"); return MNC_ERR_NOMEM; }
			memcpy(fq->bases.p + nb, p, l);
			nb += (int64_t)l;
		}
		const int64_t s_len = nb - s_off;
		if (memchr(fq->bases.p + s_off, ' ', (size_t)s_len) || memchr(fq->bases.p + s_off, '\t', (size_t)s_len))
			return fq_fail("Whitespace is not allowed in the sequence.");
		// ---- quality lines: until a line starting with '@' once enough characters are held
		if (!fq->quals.ensure((size_t)nb + 64, (size_t)s_off)) { set_error("out of host memory"); return MNC_ERR_NOMEM; }
		int64_t q_len = 0;
		bool q_first = true;
		for (;;) {
			if (!fq->in.next(p, len)) {
				if (fq->in.err) return fq_io_fail(fq);
				fq->done = true;
				break;
			}
			if (!q_first && len > 0 && p[0] == '@' && q_len >= s_len) {
				fq->pending.assign(p, len);
				fq->have_pending = true;
				break;
			}
			q_first = false;
			const size_t l = rstrip_len(p, len);
			if (!fq->quals.ensure((size_t)(s_off + q_len) + l + 64, (size_t)(s_off + q_len))) { set_error("out of host memory"); return MNC_ERR_NOMEM; }
			memcpy(fq->quals.p + s_off + q_len, p, l);
			q_len += (int64_t)l;
		}
		if (q_len != s_len) {
			set_error("Lengths of sequence and quality values differs for %s (%lld and %lld).",
			          fq->titles.substr(t_off).c_str(), (long long)s_len, (long long)q_len);
			return MNC_ERR_FORMAT;
		}
		for (int64_t i = 0; i < q_len; ++i) {
			const uint8_t c = fq->quals.p[s_off + i];
			if (c < 33 || c > 126) return fq_fail("Invalid character in quality string");
		}
		// ---- record complete
		uint32_t idl = 0, ido = 0;
		{
			const char *t = fq->titles.data() + t_off;
			size_t a = 0;
			while (a < t_len && (t[a] == ' ' || (t[a] >= 9 && t[a] <= 13))) ++a;
			size_t b = a;
			while (b < t_len && !(t[b] == ' ' || (t[b] >= 9 && t[b] <= 13))) ++b;
			idl = (uint32_t)(b - a);
			ido = (uint32_t)a;
		}
		fq->id_len.push_back(idl);
		fq->id_off.push_back(ido);
		fq->title_off.push_back(fq->titles.size());
		fq->offsets.push_back(nb);
		++fq->n;
		if (fq->n >= max_reads || (uint64_t)nb >= max_bases) break;
	}
	if (!fq->have_pending) fq->done = true;
	if (fq->n > 0 && !fq->bases.p) {       // all-empty reads: keep the accessors non-NULL
		if (!fq->bases.ensure(64, 0) || !fq->quals.ensure(64, 0)) { set_error("out of host memory"); return MNC_ERR_NOMEM; }
	}
	*n_reads = fq->n;
	return MNC_OK;
}

extern "C" const uint8_t *mnc_fastq_bases(const mnc_fastq *fq) { return fq ? fq->bases.p : nullptr; }
extern "C" const int64_t *mnc_fastq_offsets(const mnc_fastq *fq) { return fq ? fq->offsets.data() : nullptr; }
extern "C" const uint8_t *mnc_fastq_quals(const mnc_fastq *fq) { return fq ? fq->quals.p : nullptr; }

extern "C" int mnc_fastq_title(const mnc_fastq *fq, uint32_t r, const char **title, uint32_t *len, uint32_t *id_len)
{
	if (!fq || r >= fq->n) return MNC_ERR_ARG;
	if (title) *title = fq->titles.data() + fq->title_off[r];
	if (len) *len = (uint32_t)(fq->title_off[r + 1] - fq->title_off[r]);
	if (id_len) *id_len = fq->id_len[r];
	return MNC_OK;
}

namespace {
struct OutFile {
	FILE *f = nullptr;
	std::string buf;
	int flush()
	{
		if (f && !buf.empty()) {
			if (fwrite(buf.data(), 1, buf.size(), f) != buf.size()) return MNC_ERR_IO;
			buf.clear();
		}
		return MNC_OK;
	}
};
}

extern "C" int mnc_fastq_route(const mnc_fastq *fq, const uint8_t *dest, const int32_t *label,
                               const char *const *labels, int n_labels, const char *const *paths)
{
	if (!fq || !dest || !paths) return MNC_ERR_ARG;
	OutFile out[4];
	int rc = MNC_OK;
	for (uint32_t r = 0; r < fq->n && rc == MNC_OK; ++r) {
		const uint8_t d = dest[r];
		if (!d) continue;
		const char *t = fq->titles.data() + fq->title_off[r];
		const size_t t_len = (size_t)(fq->title_off[r + 1] - fq->title_off[r]);
		const int64_t o = fq->offsets[r], l = fq->offsets[r + 1] - o;
		for (int k = 0; k < 4; ++k) {
			if (!(d >> k & 1)) continue;
			OutFile &of = out[k];
			if (!of.f) {
				if (!paths[k]) { set_error("read %u is routed to a file that was not given", r); rc = MNC_ERR_ARG; break; }
				of.f = fopen(paths[k], "ab");
				if (!of.f) { set_error("cannot open %s: %s", paths[k], strerror(errno)); rc = MNC_ERR_IO; break; }
			}
			std::string &b = of.buf;
			b.push_back('@');
			if (k == 2 && label && label[r] >= 0) {           // seq_record.id = tax_unit
				if (!labels || label[r] >= n_labels) { set_error("label %d of read %u is out of range", label[r], r); rc = MNC_ERR_ARG; break; }
				const char *id = labels[label[r]];
				const size_t idl = strlen(id);
				if (t_len == 0) b.append(id, idl);
				else if (fq->id_len[r] == idl && memcmp(t + fq->id_off[r], id, idl) == 0) b.append(t, t_len);
				else { b.append(id, idl); b.push_back(' '); b.append(t, t_len); }
			} else b.append(t, t_len);
			b.push_back('\n');
			b.append((const char*)fq->bases.p + o, (size_t)l);
			b.append("\n+\n", 3);
			b.append((const char*)fq->quals.p + o, (size_t)l);
			b.push_back('\n');
			if (b.size() > (4u << 20)) { rc = of.flush(); if (rc) { set_error("write to %s failed", paths[k]); break; } }
		}
	}
	for (int k = 0; k < 4; ++k) {
		if (!out[k].f) continue;
		if (rc == MNC_OK && out[k].flush() != MNC_OK) { set_error("write to %s failed", paths[k]); rc = MNC_ERR_IO; }
		if (fclose(out[k].f) != 0 && rc == MNC_OK) { set_error("write to %s failed", paths[k]); rc = MNC_ERR_IO; }
	}
	return rc;
}

// ---------------------------------------------------------------- hits carried across index parts
namespace {
struct HitState { int32_t hits, nm, mlen, name, tied; };
}

struct mnc_hitmap {
	std::unordered_map<std::string, HitState> m;
	std::vector<std::string> names;
	std::unordered_map<std::string, int> name_id;
	int intern(const std::string &s)
	{
		auto it = name_id.find(s);
		if (it != name_id.end()) return it->second;
		const int id = (int)names.size();
		names.push_back(s);
		name_id.emplace(s, id);
		return id;
	}
};

extern "C" int mnc_hitmap_create(mnc_hitmap **out)
{
	if (!out) return MNC_ERR_ARG;
	*out = new mnc_hitmap();
	return MNC_OK;
}

extern "C" void mnc_hitmap_free(mnc_hitmap *hm) { delete hm; }
extern "C" int64_t mnc_hitmap_size(const mnc_hitmap *hm) { return hm ? (int64_t)hm->m.size() : 0; }
extern "C" int mnc_hitmap_n_names(const mnc_hitmap *hm) { return hm ? (int)hm->names.size() : 0; }
extern "C" const char *mnc_hitmap_name(const mnc_hitmap *hm, int id)
{
	return (hm && id >= 0 && id < (int)hm->names.size()) ? hm->names[id].c_str() : nullptr;
}

static const char HITMAP_MAGIC[8] = {'M', 'N', 'C', 'H', 'I', 'T', 'S', '1'};

extern "C" int mnc_hitmap_save(const mnc_hitmap *hm, const char *path)
{
	if (!hm || !path) return MNC_ERR_ARG;
	FILE *f = fopen(path, "wb");
	if (!f) { set_error("cannot open %s: %s", path, strerror(errno)); return MNC_ERR_IO; }
	bool ok = fwrite(HITMAP_MAGIC, 1, 8, f) == 8;
	auto put_u64 = [&](uint64_t v) { ok = ok && fwrite(&v, 8, 1, f) == 1; };
	auto put_str = [&](const std::string &s) { put_u64(s.size()); ok = ok && (s.empty() || fwrite(s.data(), 1, s.size(), f) == s.size()); };
	put_u64(hm->names.size());
	for (const std::string &s : hm->names) put_str(s);
	put_u64(hm->m.size());
	for (const auto &kv : hm->m) {
		put_str(kv.first);
		ok = ok && fwrite(&kv.second, sizeof(HitState), 1, f) == 1;
	}
	if (fclose(f) != 0) ok = false;
	if (!ok) { set_error("write to %s failed", path); return MNC_ERR_IO; }
	return MNC_OK;
}

extern "C" int mnc_hitmap_load(const char *path, mnc_hitmap **out)
{
	if (!path || !out) return MNC_ERR_ARG;
	FILE *f = fopen(path, "rb");
	if (!f) { set_error("cannot open %s: %s", path, strerror(errno)); return MNC_ERR_IO; }
	mnc_hitmap *hm = new mnc_hitmap();
	bool ok = true;
	char magic[8];
	ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, HITMAP_MAGIC, 8) == 0;
	auto get_u64 = [&](uint64_t &v) { ok = ok && fread(&v, 8, 1, f) == 1; };
	auto get_str = [&](std::string &s) {
		uint64_t l = 0;
		get_u64(l);
		if (!ok || l > (1u << 24)) { ok = false; return; }
		s.resize(l);
		ok = l == 0 || fread(&s[0], 1, l, f) == l;
	};
	uint64_t nn = 0, ne = 0;
	get_u64(nn);
	for (uint64_t i = 0; ok && i < nn; ++i) { std::string s; get_str(s); if (ok) hm->intern(s); }
	get_u64(ne);
	for (uint64_t i = 0; ok && i < ne; ++i) {
		std::string s;
		HitState st;
		get_str(s);
		ok = ok && fread(&st, sizeof(HitState), 1, f) == 1 && st.name >= 0 && st.name < (int)hm->names.size();
		if (ok) hm->m.emplace(std::move(s), st);
	}
	fclose(f);
	if (!ok) { delete hm; set_error("%s is not a carried-hits file", path); return MNC_ERR_FORMAT; }
	*out = hm;
	return MNC_OK;
}

extern "C" int mnc_hitmap_update(mnc_hitmap *hm, const mnc_fastq *fq, const mnc_index *idx,
                                 const int32_t *assign, const mnc_hit_t *best, const int32_t *nhits, int32_t *out)
{
	if (!hm || !fq || !idx || !assign || !best || !nhits || !out) return MNC_ERR_ARG;
	std::vector<int> name_of(idx->contig_name.size(), -1);
	std::string id;
	for (uint32_t r = 0; r < fq->n; ++r) {
		id.assign(fq->titles.data() + fq->title_off[r] + fq->id_off[r], fq->id_len[r]);
		HitState *st = nullptr;
		if (nhits[r] > 0) {
			const mnc_hit_t &b = best[r];
			if (b.rid < 0 || b.rid >= (int32_t)name_of.size()) { set_error("read %u: contig %d is not in the index", r, b.rid); return MNC_ERR_ARG; }
			if (name_of[b.rid] < 0) name_of[b.rid] = hm->intern(idx->contig_name[b.rid]);
			const int32_t tied = assign[r] == MNC_AMBIGUOUS;
			auto ins = hm->m.emplace(id, HitState{0, 0, 1, -1, 0});
			st = &ins.first->second;
			if (st->hits == 0) {
				*st = HitState{nhits[r], b.nm, b.mlen, name_of[b.rid], tied};
			} else {
				const long long l = (long long)b.nm * st->mlen, rr = (long long)st->nm * b.mlen;
				if (l < rr) { st->nm = b.nm, st->mlen = b.mlen, st->name = name_of[b.rid], st->tied = tied; }
				else if (l == rr) { st->nm = b.nm, st->mlen = b.mlen, st->name = name_of[b.rid], st->tied = 1; }
				st->hits += nhits[r];
			}
		} else {
			auto it = hm->m.find(id);
			if (it != hm->m.end()) st = &it->second;
		}
		int32_t *o = out + (size_t)r * 5;
		if (st) o[0] = st->hits, o[1] = st->nm, o[2] = st->mlen, o[3] = st->name, o[4] = st->tied;
		else o[0] = o[1] = o[2] = o[4] = 0, o[3] = -1;
	}
	return MNC_OK;
}
