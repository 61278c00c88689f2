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
#include <sys/uio.h>
#include <sys/mman.h>
#include <sys/statvfs.h>
#include <climits>
#ifndef IOV_MAX
#define IOV_MAX 1024
#endif
#include <algorithm>
#include <unistd.h>
#include <sched.h>
#include "team.h"
#include <mutex>
#include <atomic>
#include <vector>
#include <string>
#include <unordered_map>

using namespace mnc;

namespace {

constexpr size_t IO_CHUNK = 8u << 20;
// sample files being worked on side by side (mnc_host_set_io_workers): the parse / routing teams share the cores
std::atomic<int> g_io_workers{1};

// Giving a gigabyte back to the system and asking for it again per file costs ~0.1 s of page work each
// way: the readers' large buffers are kept (up to CACHE_KEEP bytes) and handed to the next reader.
struct BigCache {
	std::mutex mu;
	std::vector<std::pair<char*, size_t>> kept;
	size_t kept_bytes = 0;
	static constexpr size_t CACHE_KEEP = 6ull << 30, BIG = 16u << 20;
	// `most`: no kept buffer larger than this (0: any) -- a reader that leaves a shared block takes a block of the size it
	// needs, not a kept gigabyte that would then be pinned by the next small batch
	char *take(size_t need, size_t *got, size_t most = 0)
	{
		std::lock_guard<std::mutex> g(mu);
		size_t best = kept.size();
		for (size_t i = 0; i < kept.size(); ++i)
			if (kept[i].second >= need && (most == 0 || kept[i].second <= most) && (best == kept.size() || kept[i].second < kept[best].second)) best = i;
		if (best == kept.size()) return nullptr;
		char *p = kept[best].first;
		*got = kept[best].second;
		kept_bytes -= kept[best].second;
		kept[best] = kept.back(), kept.pop_back();
		return p;
	}
	void give(char *p, size_t bytes)
	{
		if (!p) return;
		{
			std::lock_guard<std::mutex> g(mu);
			if (bytes >= BIG && kept_bytes + bytes <= CACHE_KEEP) { kept.push_back({ p, bytes }), kept_bytes += bytes; return; }
		}
		free(p);
	}
};
BigCache g_big;

// A block of file text that a reader and the batches it has handed out hold together: the batches' records lie in it
// (written out of it by the routing pass), the reader goes on parsing behind them.  Whoever lets go last gives it back.
struct SharedBuf {
	char *p;
	size_t cap;
	std::atomic<int> refs;
};
void shared_drop(SharedBuf *sb)
{
	if (sb && sb->refs.fetch_sub(1) == 1) { g_big.give(sb->p, sb->cap); delete sb; }
}

struct LineReader {
	int fd = -1;
	char *data = nullptr;       // raw buffer (no zero-filling when it grows: a batch holds a gigabyte)
	size_t cap = 0;
	size_t lo = 0, hi = 0;
	bool eof = false, seekable = false;
	int err = 0;                // errno of a failed read(): reported as MNC_ERR_IO, never taken for the end of the file

	bool grow(size_t need)
	{
		if (need <= cap) return true;
		size_t nc = cap + cap / 2;
		if (nc < need) nc = need;
		size_t got = 0;
		if (char *c = g_big.take(nc, &got)) {                      // a kept buffer: move what is held
			if (hi > lo) memcpy(c + lo, data + lo, hi - lo);
			g_big.give(data, cap);
			data = c, cap = got;
			return true;
		}
		char *np_ = (char*)realloc(data, nc);
		if (!np_) return false;
		data = np_, cap = nc;
		return true;
	}
	void release() { g_big.give(data, cap); data = nullptr, cap = 0, lo = hi = 0; }
	void compact() { if (lo > 0) { memmove(data, data + lo, hi - lo); hi -= lo, lo = 0; } }

	// next line without its terminator; false at end of file.  `raw_len` = bytes consumed
	bool next(const char *&p, size_t &len)
	{
		for (;;) {
			const char *nl = hi > lo ? (const char*)memchr(data + lo, '\n', hi - lo) : nullptr;
			if (nl) {
				p = data + lo;
				len = (size_t)(nl - p);
				lo += len + 1;
				return true;
			}
			if (eof) {
				if (hi == lo) return false;
				p = data + lo, len = hi - lo, lo = hi;
				return true;
			}
			// refill: keep the partial line, append a chunk
			compact();
			if (!grow(hi + IO_CHUNK)) { err = ENOMEM; eof = true; return false; }
			ssize_t n;
			do n = read(fd, data + hi, IO_CHUNK); while (n < 0 && errno == EINTR);
			if (n < 0) { err = errno ? errno : EIO; eof = true; return false; }
			if (n == 0) eof = true;
			hi += (size_t)n;
		}
	}

	// have `want` bytes after `lo` (or everything up to the end of the file): slices of a regular file
	// are read by several threads at once
	bool fill(size_t want)
	{
		if (hi - lo >= want || eof) return true;
		compact();
		if (!grow(want + IO_CHUNK)) { err = ENOMEM; return false; }
		const off_t pos = seekable ? lseek(fd, 0, SEEK_CUR) : (off_t)-1;
		struct stat sb;
		if (pos >= 0 && fstat(fd, &sb) == 0 && sb.st_size > pos) {
			const size_t n = std::min<size_t>(want - hi, (size_t)(sb.st_size - pos));
			const size_t slice = 16u << 20;
			const int64_t n_slices = (int64_t)((n + slice - 1) / slice);
			int fail = 0;
			std::atomic<int64_t> next_slice{0};
			team().run((int)std::min<int64_t>(io_threads(), n_slices), [&](int, int) {
			for (int64_t k; (k = next_slice.fetch_add(1, std::memory_order_relaxed)) < n_slices;) {
				size_t done = 0;
				const size_t len = std::min(slice, n - (size_t)k * slice);
				while (done < len) {
					const ssize_t r = pread(fd, data + hi + (size_t)k * slice + done, len - done, pos + (off_t)((size_t)k * slice + done));
					if (r < 0 && errno == EINTR) continue;
					if (r <= 0) {
						__atomic_store_n(&fail, r < 0 ? (errno ? errno : EIO) : EIO, __ATOMIC_RELAXED);
						break;
					}
					done += (size_t)r;
				}
			}
			});
			if (fail) { err = fail; eof = true; return false; }
			hi += n;
			if (lseek(fd, pos + (off_t)n, SEEK_SET) < 0) { err = errno ? errno : EIO; eof = true; return false; }
		}
		while (hi - lo < want && !eof) {                          // the rest (or all of it, from a pipe)
			if (!grow(hi + IO_CHUNK)) { err = ENOMEM; return false; }
			ssize_t n;
			do n = read(fd, data + hi, std::min(IO_CHUNK, want - (hi - lo) + 1)); while (n < 0 && errno == EINTR);
			if (n < 0) { err = errno ? errno : EIO; eof = true; return false; }
			if (n == 0) eof = true;
			hi += (size_t)n;
		}
		return true;
	}
	// Host threads of one parse or routing pass (MNC_IO_THREADS caps it: the aligner's pipeline runs a parse and a
	// routing pass side by side with the thread that launches kernels).  A host that works on several sample files at
	// once (monica's ThreadPool, aligner.py:89-103: one sample per worker) says so with mnc_host_set_io_workers: the
	// cores are shared out over the workers, or W workers x 2 passes x 16 threads oversubscribe a large GPU host.
	// the host threads this process may keep busy: the hardware's, or the share its control group is given of them
	// (cpu.max: the GPU box shows 256 hardware threads and grants 16 -- a team of 256 would only queue up for them)
	static int host_cores()
	{
		static const int n = [] {
			int hw = (int)std::thread::hardware_concurrency();
			if (hw < 1) hw = 1;
			cpu_set_t set;                                        // (a process started under taskset / numactl)
			if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int a = CPU_COUNT(&set); if (a >= 1 && a < hw) hw = a; }
			long long quota = 0, period = 0;
			bool have = false;
			if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                     // control groups v2: "<quota | max> <period>"
				have = fscanf(f, "%lld %lld", &quota, &period) == 2;
				fclose(f);
			} else {                                                                   // v1
				FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"), *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
				if (fq && fp) have = fscanf(fq, "%lld", &quota) == 1 && fscanf(fp, "%lld", &period) == 1;
				if (fq) fclose(fq);
				if (fp) fclose(fp);
			}
			if (have && quota > 0 && period > 0) {
				const int share = (int)((quota + period - 1) / period);
				if (share >= 1 && share < hw) hw = share;
			}
			return hw;
		}();
		return n;
	}
	static int io_threads()
	{
		static const int cap = [] { const char *e = getenv("MNC_IO_THREADS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 16; }();
		int t = host_cores() / std::max(1, g_io_workers.load(std::memory_order_relaxed));
		return t < 1 ? 1 : t > cap ? cap : t;
	}
	// The aligner's pipeline runs a parse and a routing pass side by side: each may be given a team of its own size
	// (MNC_PARSE_THREADS / MNC_ROUTE_THREADS; default: io_threads() each -- on the GPU box both passes still speed up
	// from 6 to 16 threads while they run side by side, profiles/r04f_files_sweep.txt)
	static int team_threads(const char *env)
	{
		const char *e = getenv(env);
		const int v = e ? atoi(e) : 0;
		const int all = io_threads();
		return v > 0 ? (v < 64 ? v : 64) : all;
	}
	static int parse_threads() { return team_threads("MNC_PARSE_THREADS"); }
	static int route_threads() { return team_threads("MNC_ROUTE_THREADS"); }
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
		uint8_t *np_ = nullptr;
		if (pinned) np_ = (uint8_t*)mnc_host_alloc(nc);
		else {
			size_t got = 0;
			np_ = (uint8_t*)g_big.take(nc, &got);
			if (np_) nc = got; else np_ = (uint8_t*)malloc(nc);
		}
		if (!np_) return false;
		if (keep) memcpy(np_, p, keep);
		release();
		p = np_, cap = nc;
		return true;
	}
	void release()
	{
		if (p) { if (pinned) mnc_host_free(p); else g_big.give((char*)p, cap); }
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
	std::vector<size_t> nl;          // the four-line fast path: line ends of the buffered text
	// A batch of four-line records keeps the file's own bytes: the routing pass writes a record that is not rewritten
	// (unmapped, ambiguous, focus; the body of a mapped one) straight out of them, and the qualities are never copied.
	bool pending_in_buf = false;     // the pending title line has not been consumed: its bytes start at in.lo
	bool text_backed = false;        // this batch's records lie in `text` (the reader: in.data) at rec_off / seq_off / qual_off
	const char *text = nullptr;      // a detached batch: the reader's buffer at the time, which it holds on to through `sb`
	SharedBuf *sb = nullptr;         // batch: the block its text lies in; reader: the block in.data is, once a batch shares it
	std::vector<uint64_t> rec_off, seq_off, qual_off;   // n + 1 / n / n offsets into the text
	std::vector<uint8_t> verbatim;   // n: the record's text is byte for byte what the writer writes (no trailing blanks, a bare '+' line)
	std::atomic<bool> quals_ready{true};   // the quality array holds this batch's (a text-backed batch fills it on demand)
	std::mutex q_mu;
	const char *text_base() const { return text ? text : in.data; }
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

// A host that works on `n_workers` sample files at once (monica's ThreadPool, aligner.py:89-103): the parse / routing
// teams of every reader take cores / n_workers threads each from now on (1: a reader may use them all).
extern "C" int mnc_host_set_io_workers(int n_workers)
{
	if (n_workers < 1) return MNC_ERR_ARG;
	g_io_workers.store(n_workers, std::memory_order_relaxed);
	return MNC_OK;
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
		const bool ok = fstat(fd, &sb) == 0;
		fq->file_size = ok ? (int64_t)sb.st_size : 0;
		fq->in.seekable = ok && S_ISREG(sb.st_mode) && lseek(fd, 0, SEEK_CUR) >= 0;
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
	if (fq->sb && fq->in.data == fq->sb->p) fq->in.data = nullptr, fq->in.cap = 0;   // the reader's block is shared: let go of it, the last holder frees it
	shared_drop(fq->sb);
	fq->in.release();
	fq->bases.release();
	fq->quals.release();
	delete fq;
}

// bytes of the file that no batch has taken yet (-1: not a regular file): a host loop sizes its last batches by it
extern "C" int mnc_fastq_remaining(const mnc_fastq *fq, int64_t *bytes)
{
	if (!fq || !bytes) return MNC_ERR_ARG;
	*bytes = -1;
	if (fq->in.fd < 0 || !fq->in.seekable) return MNC_OK;
	const off_t pos = lseek(fq->in.fd, 0, SEEK_CUR);
	if (pos < 0) return MNC_OK;
	const int64_t buffered = (int64_t)(fq->in.hi - fq->in.lo) + (fq->have_pending && !fq->pending_in_buf ? (int64_t)fq->pending.size() + 1 : 0);
	*bytes = fq->done ? 0 : std::max<int64_t>(0, fq->file_size - (int64_t)pos) + buffered;
	return MNC_OK;
}

// The current batch leaves the reader: a second handle (no file behind it) takes the batch's arrays, the
// reader starts its next batch in fresh ones.  A host loop can then parse batch k + 1 while batch k is being
// classified and batch k - 1 written to the routing folders; every accessor, mnc_fastq_route and
// mnc_hitmap_update work on the detached handle as on the reader; mnc_fastq_close frees it.
extern "C" int mnc_fastq_detach_batch(mnc_fastq *fq, mnc_fastq **out)
{
	if (!fq || !out) return MNC_ERR_ARG;
	mnc_fastq *b = new (std::nothrow) mnc_fastq();
	if (!b) return MNC_ERR_NOMEM;
	if (fq->text_backed && !fq->sb) {
		fq->sb = new (std::nothrow) SharedBuf{ fq->in.data, fq->in.cap, { 1 } };
		if (!fq->sb) { delete b; return MNC_ERR_NOMEM; }
	}
	b->in.fd = -1;
	b->done = true;
	b->bases.pinned = true;
	b->n = fq->n;
	std::swap(b->bases, fq->bases), std::swap(b->quals, fq->quals);
	b->offsets.swap(fq->offsets), b->titles.swap(fq->titles), b->title_off.swap(fq->title_off);
	b->id_len.swap(fq->id_len), b->id_off.swap(fq->id_off);
	b->quals_ready.store(fq->quals_ready.load()), fq->quals_ready.store(true);
	if (fq->text_backed) {
		// the batch and the reader share the buffer from here on: the batch's records lie in it, the reader parses on behind
		// them; nothing is copied until the reader has to move its data (fastq_unshare)
		LineReader &in = fq->in;
		(void)in;
		fq->sb->refs.fetch_add(1);
		b->sb = fq->sb, b->text = in.data, b->text_backed = true;
		b->rec_off.swap(fq->rec_off), b->seq_off.swap(fq->seq_off), b->qual_off.swap(fq->qual_off), b->verbatim.swap(fq->verbatim);
		fq->text_backed = false;
	}
	fq->bases.pinned = true;
	fq->n = 0;
	fq->offsets.assign(1, 0), fq->title_off.assign(1, 0);
	*out = b;
	return MNC_OK;
}

// ---------------------------------------------------------------- the four-line fast path
// A batch of records that each take exactly four lines (title, sequence, '+' line, qualities -- what
// every basecaller writes) is found and copied by all host threads at once: line ends by slices of
// the buffered text, then per record the checks under which the general parser below (Biopython's
// FastqGeneralIterator, which allows wrapped sequences) reads the same four lines the same way --
// '@' first, '+' third with an empty or equal caption, equal lengths, a next line that starts
// with '@'.  Anything else -- including what the general parser reports as an error -- leaves the
// state untouched and is parsed (or reported) by the general parser.
// The reader is about to move or grow its buffer while batches still hold records in it: it goes on in a block of its
// own with what it has not consumed (what LineReader::compact would have moved anyway), the batches keep the old one.
static bool fastq_unshare(mnc_fastq *fq, size_t room)
{
	if (!fq->sb) return true;
	LineReader &in = fq->in;
	if (fq->sb->refs.load() == 1) {                               // every batch has let go: the block is the reader's alone again
		delete fq->sb;
		fq->sb = nullptr;
		return true;
	}
	const size_t tail = in.hi - in.lo;
	size_t got = 0, need = std::max<size_t>(tail + IO_CHUNK, room);
	// (the tail plus the next refill, or what the caller is about to read: every detached batch keeps the block it was
	// parsed in alive, so a block is sized by what it will hold, and at most three batches are in flight per sample --
	// parsed, on the GPU, being written: monica_amd/aligner.py)
	char *nb = g_big.take(need, &got, 2 * need + (64u << 20));
	if (!nb) { nb = (char*)malloc(need); got = need; }
	if (!nb) return false;
	if (tail) memcpy(nb, in.data + in.lo, tail);
	shared_drop(fq->sb);
	fq->sb = nullptr;
	in.data = nb, in.cap = got, in.lo = 0, in.hi = tail;
	return true;
}

static int fastq_next_fast(mnc_fastq *fq, uint32_t max_reads, uint64_t max_bases, bool *handled)
{
	*handled = false;
	LineReader &in = fq->in;
	// (the pending title line must still lie in the buffer, in front of its record: the batch keeps the records' own bytes)
	if (!fq->have_pending || !fq->pending_in_buf || !in.seekable || max_reads == 0) return MNC_OK;
	{
		uint64_t want = 2 * max_bases + (uint64_t)max_reads * 512 + (1u << 20);
		const uint64_t left = (uint64_t)fq->file_size + (1u << 20);       // never more than the file
		if (want > left) want = left;
		if (!(in.hi - in.lo >= want || in.eof) && !fastq_unshare(fq, (size_t)want + IO_CHUNK)) { set_error("out of host memory"); return MNC_ERR_NOMEM; }   // fill() is going to move the data
		if (!in.fill((size_t)want)) return in.err == ENOMEM ? (set_error("out of host memory"), MNC_ERR_NOMEM) : MNC_OK;   // an I/O error: the general parser reports it
	}
	const char *base = in.data + in.lo;
	const size_t avail = in.hi - in.lo;
	if (avail == 0) return MNC_OK;
	const int T = LineReader::parse_threads();
	// ---- line ends
	std::vector<std::vector<size_t>> part((size_t)T);
	team().run(T, [&](int t, int nt) {
		const size_t a = avail * (size_t)t / (size_t)nt, b = avail * (size_t)(t + 1) / (size_t)nt;
		std::vector<size_t> &v = part[(size_t)t];
		v.reserve((b - a) / 1024 + 16);
		const char *p = base + a, *e = base + b;
		while (p < e) {
			const char *q = (const char*)memchr(p, '\n', (size_t)(e - p));
			if (!q) break;
			v.push_back((size_t)(q - base));
			p = q + 1;
		}
	});
	std::vector<size_t> &nl = fq->nl;
	nl.clear();
	for (const auto &v : part) nl.insert(nl.end(), v.begin(), v.end());
	bool open_end = false;                                        // the file ends without a line terminator
	if (in.eof && (nl.empty() || nl.back() != avail - 1)) nl.push_back(avail), open_end = true;
	const size_t n_lines = nl.size();
	auto line = [&](size_t i, const char *&p, size_t &len) { const size_t s0 = i ? nl[i - 1] + 1 : 0; p = base + s0, len = nl[i] - s0; };
	// lines of the buffer: title0 seq0 plus0 qual0 title1 seq1 ...
	size_t R;
	if (in.eof) {
		if (n_lines % 4 != 0) return MNC_OK;                      // not whole four-line records to the end
		R = n_lines / 4;
	} else R = n_lines ? (n_lines - 1) / 4 : 0;                   // only records whose next title is in the buffer
	if (R > max_reads) R = max_reads;
	if (R == 0) return MNC_OK;
	std::vector<uint32_t> slen(R), tlen(R);
	fq->verbatim.resize(R);
	std::atomic<int> bad_any{0};
	team().slices(T, (int64_t)R, [&](int64_t r) {
		const char *tp, *sp, *pp, *qp;
		size_t tl, sl, pl, ql;
		line((size_t)(4 * r), tp, tl), line((size_t)(4 * r + 1), sp, sl), line((size_t)(4 * r + 2), pp, pl), line((size_t)(4 * r + 3), qp, ql);
		if (tl == 0 || tp[0] != '@' || pl == 0 || pp[0] != '+') { bad_any.store(1, std::memory_order_relaxed); return; }
		const size_t t_len = rstrip_len(tp + 1, tl - 1), c_len = rstrip_len(pp + 1, pl - 1);
		if (c_len > 0 && (c_len != t_len || memcmp(pp + 1, tp + 1, t_len) != 0)) { bad_any.store(1, std::memory_order_relaxed); return; }
		const size_t s_len = rstrip_len(sp, sl), q_len = rstrip_len(qp, ql);
		if (s_len != q_len || s_len > 0xffffffffu) { bad_any.store(1, std::memory_order_relaxed); return; }
		if (memchr(sp, ' ', s_len) || memchr(sp, '\t', s_len)) { bad_any.store(1, std::memory_order_relaxed); return; }
		unsigned out_of_range = 0;                                // (no early exit: the loop is a vector OR)
		for (size_t i = 0; i < q_len; ++i) out_of_range |= (unsigned)((uint8_t)(qp[i] - 33) > 93);
		if (out_of_range) { bad_any.store(1, std::memory_order_relaxed); return; }
		if ((size_t)(4 * r + 4) < n_lines) {
			const char *np_; size_t nl_;
			line((size_t)(4 * r + 4), np_, nl_);
			if (nl_ == 0 || np_[0] != '@') bad_any.store(1, std::memory_order_relaxed);
		}
		slen[(size_t)r] = (uint32_t)s_len, tlen[(size_t)r] = (uint32_t)t_len;
		// the record's bytes are what FastqPhredWriter writes for it: nothing stripped, a bare '+', a line end behind the qualities
		fq->verbatim[(size_t)r] = (t_len == tl - 1 && pl == 1 && s_len == sl && q_len == ql && !(open_end && (size_t)(4 * r + 3) == n_lines - 1)) ? 1 : 0;
	});
	if (bad_any.load()) return MNC_OK;
	// ---- how many records: the batch ends with the record that reaches max_bases
	size_t n = 0;
	{
		uint64_t nb = 0;
		while (n < R) { nb += slen[n]; ++n; if (nb >= max_bases) break; }
	}
	fq->offsets.resize(n + 1), fq->title_off.resize(n + 1), fq->id_len.resize(n), fq->id_off.resize(n);
	fq->rec_off.resize(n + 1), fq->seq_off.resize(n), fq->qual_off.resize(n), fq->verbatim.resize(n);
	fq->offsets[0] = 0, fq->title_off[0] = 0;
	for (size_t r = 0; r < n; ++r) {
		fq->offsets[r + 1] = fq->offsets[r] + slen[r];
		fq->title_off[r + 1] = fq->title_off[r] + tlen[r];
	}
	const int64_t nb = fq->offsets[n];
	if (!fq->bases.ensure((size_t)nb + 64, 0)) { set_error("out of host memory"); return MNC_ERR_NOMEM; }
	fq->titles.resize((size_t)fq->title_off[n]);
	const size_t text0 = in.lo;                                   // offsets into in.data (the batch's text)
	team().slices(T, (int64_t)n, [&](int64_t r) {
		const char *tp, *sp, *qp;
		size_t tl, sl, ql;
		line((size_t)(4 * r), tp, tl), line((size_t)(4 * r + 1), sp, sl), line((size_t)(4 * r + 3), qp, ql);
		(void)tl, (void)sl, (void)ql;
		const size_t s_len = slen[(size_t)r], t_len = tlen[(size_t)r];
		memcpy(fq->bases.p + fq->offsets[(size_t)r], sp, s_len);
		fq->rec_off[(size_t)r] = text0 + (size_t)(tp - base), fq->seq_off[(size_t)r] = text0 + (size_t)(sp - base), fq->qual_off[(size_t)r] = text0 + (size_t)(qp - base);
		char *t = &fq->titles[(size_t)fq->title_off[(size_t)r]];
		memcpy(t, tp + 1, t_len);
		size_t a = 0;
		while (a < t_len && (t[a] == ' ' || (t[a] >= 9 && t[a] <= 13))) ++a;
		size_t b = a;
		while (b < t_len && !(t[b] == ' ' || (t[b] >= 9 && t[b] <= 13))) ++b;
		fq->id_len[(size_t)r] = (uint32_t)(b - a), fq->id_off[(size_t)r] = (uint32_t)a;
	});
	fq->n = (uint32_t)n;
	fq->text_backed = true, fq->quals_ready = false;
	// ---- what was consumed: the records; the next record's title line stays in the buffer (the pending line)
	const size_t end = nl[4 * n - 1];
	const bool last = 4 * n == n_lines;
	in.lo += (open_end && last) ? end : end + 1;
	fq->rec_off[n] = in.lo;
	if (!last) fq->have_pending = true, fq->pending_in_buf = true;
	else { fq->have_pending = false, fq->pending_in_buf = false; if (in.eof) fq->done = true; }
	*handled = true;
	return MNC_OK;
}

// the quality characters of a text-backed batch, copied out of its text when somebody asks for them
static bool fastq_fill_quals(mnc_fastq *fq)
{
	std::lock_guard<std::mutex> g(fq->q_mu);
	if (fq->quals_ready) return true;
	const size_t n = fq->n;
	if (!fq->quals.ensure((size_t)fq->offsets[n] + 64, 0)) return false;
	const char *text = fq->text_base();
	const int T = LineReader::parse_threads();
	team().slices(T, (int64_t)n, [&](int64_t r) {
		memcpy(fq->quals.p + fq->offsets[(size_t)r], text + fq->qual_off[(size_t)r], (size_t)(fq->offsets[(size_t)r + 1] - fq->offsets[(size_t)r]));
	});
	fq->quals_ready = true;
	return true;
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
	fq->text_backed = false, fq->quals_ready = true;
	fq->rec_off.clear(), fq->seq_off.clear(), fq->qual_off.clear(), fq->verbatim.clear();
	if (fq->done || max_reads == 0) return MNC_OK;
	const char *p;
	size_t len;
	if (!fq->started) {
		fq->started = true;
		if (fq->in.seekable && fq->file_size > 0) fq->have_pending = true, fq->pending_in_buf = true;   // the first title line: still in the file
		else {
			if (!fq->in.next(p, len)) {
				if (fq->in.err) return fq_io_fail(fq);
				fq->done = true;                                               // empty file
				return MNC_OK;
			}
			fq->pending.assign(p, len);
			fq->have_pending = true;
		}
	}
	{
		bool handled = false;
		if (int rc = fastq_next_fast(fq, max_reads, max_bases, &handled)) return rc;
		if (handled) { *n_reads = fq->n; return MNC_OK; }
	}
	if (!fastq_unshare(fq, 0)) { set_error("out of host memory"); return MNC_ERR_NOMEM; }   // the general parser refills (and moves) the buffer as it goes
	if (fq->have_pending && fq->pending_in_buf) {                           // the general parser holds the title line as a string
		fq->pending_in_buf = false;
		if (!fq->in.next(p, len)) {
			if (fq->in.err) return fq_io_fail(fq);
			fq->have_pending = false, fq->done = true;
			return MNC_OK;
		}
		fq->pending.assign(p, len);
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
extern "C" const uint8_t *mnc_fastq_quals(const mnc_fastq *fq)
{
	if (!fq) return nullptr;
	if (!fq->quals_ready && !fastq_fill_quals(const_cast<mnc_fastq*>(fq))) return nullptr;
	return fq->quals.p;
}

extern "C" int mnc_fastq_title(const mnc_fastq *fq, uint32_t r, const char **title, uint32_t *len, uint32_t *id_len)
{
	if (!fq || r >= fq->n) return MNC_ERR_ARG;
	if (title) *title = fq->titles.data() + fq->title_off[r];
	if (len) *len = (uint32_t)(fq->title_off[r + 1] - fq->title_off[r]);
	if (id_len) *id_len = fq->id_len[r];
	return MNC_OK;
}

// A batch that still holds the file's bytes (four-line records, fastq_next_fast): a record that goes out as it came in --
// unmapped, ambiguous, a focus copy, a mapped one whose id already is its label -- is written from those bytes, runs of
// consecutive such records as one piece; a mapped record gets its new title line from a small arena and its body (sequence,
// '+', qualities: two thirds... all but ~40 bytes of it) from the text.  pwritev gathers the pieces: no record is assembled
// in memory first.  Same files, same bytes, same order as the formatting pass below.
static int route_from_text(const mnc_fastq *fq, const uint8_t *dest, const int32_t *label, const char *const *labels, const char *const *paths)
{
	const char *text = fq->text_base();
	const uint32_t n = fq->n;
	const int T = std::max(1, std::min(LineReader::route_threads(), (int)(n / 2048 + 1)));
	// title line of a mapped record: 0 = as it is, else the bytes of "@<label> " to put in front of the old title (or "@<label>" for an empty one)
	auto relabel = [&](uint32_t r, int k) -> const char* {
		if (k != 2 || !label || label[r] < 0) return nullptr;
		const char *id = labels[label[r]];
		const size_t idl = strlen(id), t_len = (size_t)(fq->title_off[r + 1] - fq->title_off[r]);
		const char *tt = fq->titles.data() + fq->title_off[r];
		if (t_len != 0 && fq->id_len[r] == idl && memcmp(tt + fq->id_off[r], id, idl) == 0) return nullptr;
		return id;
	};
	auto rec_len = [&](uint32_t r, int k) -> size_t {
		const size_t t_len = (size_t)(fq->title_off[r + 1] - fq->title_off[r]), l = (size_t)(fq->offsets[r + 1] - fq->offsets[r]);
		size_t head = t_len;
		if (const char *id = relabel(r, k)) head = t_len == 0 ? strlen(id) : strlen(id) + 1 + t_len;
		return 1 + head + 1 + l + 3 + l + 1;
	};
	// bytes a thread has to make up itself (new title lines; whole records whose text is not what the writer writes)
	auto arena_len = [&](uint32_t r, int k) -> size_t {
		if (!fq->verbatim[r]) return rec_len(r, k);
		if (const char *id = relabel(r, k)) return 1 + strlen(id) + 1;
		return 0;
	};
	std::vector<size_t> slice_bytes((size_t)T * 4, 0), slice_arena((size_t)T, 0);
	team().run(T, [&](int t, int) {
		const uint32_t r0 = (uint32_t)((uint64_t)n * (uint64_t)t / (uint64_t)T), r1 = (uint32_t)((uint64_t)n * (uint64_t)(t + 1) / (uint64_t)T);
		for (uint32_t r = r0; r < r1; ++r)
			for (int k = 0; k < 4; ++k)
				if (dest[r] >> k & 1) slice_bytes[(size_t)t * 4 + k] += rec_len(r, k), slice_arena[(size_t)t] += arena_len(r, k);
	});
	int fds[4] = { -1, -1, -1, -1 };
	off_t start[4] = { 0, 0, 0, 0 };
	int rc = MNC_OK;
	for (int k = 0; k < 4 && rc == MNC_OK; ++k) {
		size_t total = 0;
		for (int t = 0; t < T; ++t) total += slice_bytes[(size_t)t * 4 + k];
		if (total == 0) continue;
		fds[k] = open(paths[k], O_WRONLY | O_CREAT, 0666);
		struct stat sb;
		if (fds[k] < 0 || fstat(fds[k], &sb) != 0) { set_error("cannot open %s: %s", paths[k], strerror(errno)); rc = MNC_ERR_IO; break; }
		start[k] = sb.st_size;                                // append: the new records follow what the file holds
	}
	// Buffered writes to ONE file take the inode's lock in turn: sixteen threads writing their slices of mapped/<sample>
	// go one at a time (measured on ext4: 0.20 s for 477 MB into one file, 0.075 s into three).  With MNC_ROUTE_MMAP=1 a
	// large append goes through a shared mapping of the file's new tail instead: the file is grown and its blocks reserved
	// once, and the threads' copies fault their pages in side by side (0.10 s on that ext4; on the GPU box's /tmp it is the
	// slower way, 0.18 against 0.12 s per GB, which is why it is not the default).  Free space is checked before: a store
	// into a mapping has no error return.  Everything else takes pwritev.
	char *map_base[4] = { nullptr, nullptr, nullptr, nullptr };
	size_t map_len[4] = { 0, 0, 0, 0 };
	off_t map_off[4] = { 0, 0, 0, 0 };
	if (rc == MNC_OK && getenv("MNC_ROUTE_MMAP") && !getenv("MNC_ROUTE_PWRITE")) {
		const long page = sysconf(_SC_PAGESIZE);
		size_t all_bytes = 0;
		for (size_t b : slice_bytes) all_bytes += b;
		for (int k = 0; k < 4; ++k) {
			size_t total = 0;
			for (int t = 0; t < T; ++t) total += slice_bytes[(size_t)t * 4 + k];
			// only the file that takes most of the batch (mapped/<sample>, as a rule): writes to different files do not wait
			// for each other, and a page of a mapping costs more than a page of a write
			if (fds[k] < 0 || total < (32u << 20) || total * 5 < all_bytes * 3) continue;
			struct statvfs vfs;
			if (fstatvfs(fds[k], &vfs) != 0 || (unsigned long long)vfs.f_bavail * vfs.f_frsize < (unsigned long long)total + (64ull << 20)) continue;
			const int fd2 = open(paths[k], O_RDWR);                // (a mapping for writing wants a descriptor that can read)
			if (fd2 < 0) continue;
			// (reserve the blocks: without that every first touch of a page allocates one under the file system's locks --
			// 0.23 s instead of 0.10 s for 477 MB on ext4 -- and a full disk would show up as a fault, not as an error)
			if (posix_fallocate(fd2, start[k], (off_t)total) != 0) { close(fd2); continue; }
			map_off[k] = start[k] / page * page;
			map_len[k] = (size_t)(start[k] - map_off[k]) + total;
			void *m = mmap(nullptr, map_len[k], PROT_READ | PROT_WRITE, MAP_SHARED, fd2, map_off[k]);
			close(fd2);
			if (m == MAP_FAILED) {                                  // back to the file's old length: pwritev appends
				map_len[k] = 0;
				if (ftruncate(fds[k], start[k]) != 0) { set_error("cannot restore %s: %s", paths[k], strerror(errno)); rc = MNC_ERR_IO; break; }
				continue;
			}
			map_base[k] = (char*)m;
		}
	}
	int fail[4] = { 0, 0, 0, 0 };
	if (rc == MNC_OK) {
		team().run(T, [&](int t, int) {
			const uint32_t r0 = (uint32_t)((uint64_t)n * (uint64_t)t / (uint64_t)T), r1 = (uint32_t)((uint64_t)n * (uint64_t)(t + 1) / (uint64_t)T);
			std::vector<char> arena(slice_arena[(size_t)t] + 16);      // sized exactly: its pieces are pointed at until they are written
			size_t used = 0;
			constexpr int IOV_FLUSH = 512;
			constexpr size_t BYTES_FLUSH = 16u << 20;
			struct Out { std::vector<iovec> iov; size_t bytes = 0; off_t pos = 0; } out[4];
			for (int k = 0; k < 4; ++k) {
				out[k].pos = start[k];
				for (int u = 0; u < t; ++u) out[k].pos += (off_t)slice_bytes[(size_t)u * 4 + k];
				out[k].iov.reserve(IOV_FLUSH + 8);
			}
			auto flush = [&](int k) {
				Out &o = out[k];
				size_t i = 0;
				off_t pos = o.pos;
				while (i < o.iov.size()) {
					const int cnt = (int)std::min<size_t>(o.iov.size() - i, IOV_MAX);
					ssize_t w = pwritev(fds[k], o.iov.data() + i, cnt, pos);
					if (w < 0 && errno == EINTR) continue;
					if (w <= 0) {
						__atomic_store_n(&fail[k], errno ? errno : EIO, __ATOMIC_RELAXED);
						break;
					}
					pos += (off_t)w;
					while (w > 0 && i < o.iov.size()) {               // a short write: go on behind what was taken
						if ((size_t)w >= o.iov[i].iov_len) { w -= (ssize_t)o.iov[i].iov_len; ++i; }
						else { o.iov[i].iov_base = (char*)o.iov[i].iov_base + w; o.iov[i].iov_len -= (size_t)w; w = 0; }
					}
				}
				o.pos += (off_t)o.bytes;
				o.iov.clear(), o.bytes = 0;
			};
			auto put = [&](int k, const char *p, size_t len) {
				Out &o = out[k];
				if (map_base[k]) {                                   // the file's new tail is mapped: copy in place
					memcpy(map_base[k] + (o.pos - map_off[k]), p, len);
					o.pos += (off_t)len;
					return;
				}
				if (!o.iov.empty() && (const char*)o.iov.back().iov_base + o.iov.back().iov_len == p) o.iov.back().iov_len += len;   // the next record of the file: one piece
				else o.iov.push_back(iovec{ (void*)p, len });
				o.bytes += len;
			};
			for (uint32_t r = r0; r < r1; ++r) {
				const uint8_t d = dest[r];
				if (!d) continue;
				const int64_t l = fq->offsets[r + 1] - fq->offsets[r];
				for (int k = 0; k < 4; ++k) {
					if (!(d >> k & 1)) continue;
					const char *id = relabel(r, k);
					if (fq->verbatim[r]) {
						if (!id) put(k, text + fq->rec_off[r], (size_t)(fq->rec_off[r + 1] - fq->rec_off[r]));
						else {
							// "@<label> " in front of the old title (the record's own '@' is skipped), or "@<label>" for an empty one
							char *h = arena.data() + used;
							const size_t idl = strlen(id), t_len = (size_t)(fq->title_off[r + 1] - fq->title_off[r]);
							h[0] = '@';
							memcpy(h + 1, id, idl);
							size_t hl = 1 + idl;
							if (t_len) h[hl++] = ' ';
							used += 1 + idl + 1;
							put(k, h, hl);
							put(k, text + fq->rec_off[r] + 1, (size_t)(fq->rec_off[r + 1] - fq->rec_off[r] - 1));
						}
					} else {
						// the record as the writer writes it, from the lines' stripped lengths
						char *h = arena.data() + used, *w = h;
						const char *tt = fq->titles.data() + fq->title_off[r];
						const size_t t_len = (size_t)(fq->title_off[r + 1] - fq->title_off[r]);
						*w++ = '@';
						if (id) { const size_t idl = strlen(id); memcpy(w, id, idl), w += idl; if (t_len) *w++ = ' '; }
						memcpy(w, tt, t_len), w += t_len;
						*w++ = '\n';
						memcpy(w, text + fq->seq_off[r], (size_t)l), w += l;
						memcpy(w, "\n+\n", 3), w += 3;
						memcpy(w, text + fq->qual_off[r], (size_t)l), w += l;
						*w++ = '\n';
						used += rec_len(r, k);
						put(k, h, (size_t)(w - h));
					}
					if (out[k].iov.size() >= (size_t)IOV_FLUSH || out[k].bytes >= BYTES_FLUSH) flush(k);
				}
			}
			for (int k = 0; k < 4; ++k) if (!out[k].iov.empty()) flush(k);
		});
	}
	for (int k = 0; k < 4; ++k) {
		if (map_base[k] && munmap(map_base[k], map_len[k]) != 0 && !fail[k]) fail[k] = errno ? errno : EIO;
		if (fds[k] < 0) continue;
		if (close(fds[k]) != 0 && !fail[k]) fail[k] = errno ? errno : EIO;
		if (rc == MNC_OK && fail[k]) { set_error("write to %s failed: %s", paths[k], strerror(fail[k])); rc = MNC_ERR_IO; }
	}
	return rc;
}

extern "C" int mnc_fastq_route(const mnc_fastq *fq, const uint8_t *dest, const int32_t *label,
                               const char *const *labels, int n_labels, const char *const *paths)
{
	if (!fq || !dest || !paths) return MNC_ERR_ARG;
	// arguments first: a destination without a path, a label out of range
	for (uint32_t r = 0; r < fq->n; ++r) {
		const uint8_t d = dest[r];
		for (int k = 0; k < 4; ++k) if ((d >> k & 1) && !paths[k]) { set_error("read %u is routed to a file that was not given", r); return MNC_ERR_ARG; }
		if ((d >> 2 & 1) && label && label[r] >= 0 && (!labels || label[r] >= n_labels)) { set_error("label %d of read %u is out of range", label[r], r); return MNC_ERR_ARG; }
	}
	// (the file's own bytes gathered by pwritev: one copy less in this process, but the kernel then reads cold memory while
	// it holds the file's lock -- on the GPU box 0.127 s per GB against 0.103 s for the pass below, which formats through
	// small buffers that are still in the cache when pwrite copies them: profiles/r04f_files_sweep.txt)
	if (fq->text_backed && fq->n > 0 && getenv("MNC_ROUTE_TEXT")) return route_from_text(fq, dest, label, labels, paths);
	const bool from_text = fq->text_backed;                      // sequence and qualities straight out of the batch's text
	if (!from_text && !fq->quals_ready && !fastq_fill_quals(const_cast<mnc_fastq*>(fq))) { set_error("out of host memory"); return MNC_ERR_NOMEM; }
	const char *text = fq->text_base();
	// Every record's length in its file is known before it is written (title, two lines of the read's
	// length, six more characters, the label in front of a mapped record's title): offsets by a prefix
	// sum, the files grown once, and every host thread formats its slice of the batch through a small
	// buffer and writes it in place (pwrite) -- the order in each file is the reads' order.
	const int T = std::max(1, std::min(LineReader::route_threads(), (int)(fq->n / 2048 + 1)));
	auto rec_len = [&](uint32_t r, int k) -> size_t {
		const size_t t_len = (size_t)(fq->title_off[r + 1] - fq->title_off[r]), l = (size_t)(fq->offsets[r + 1] - fq->offsets[r]);
		size_t head = t_len;
		if (k == 2 && label && label[r] >= 0) {
			const char *id = labels[label[r]];
			const size_t idl = strlen(id);
			const char *tt = fq->titles.data() + fq->title_off[r];
			if (t_len == 0) head = idl;
			else if (!(fq->id_len[r] == idl && memcmp(tt + fq->id_off[r], id, idl) == 0)) head = idl + 1 + t_len;
		}
		return 1 + head + 1 + l + 3 + l + 1;
	};
	std::vector<size_t> slice_bytes((size_t)T * 4, 0);
	team().run(T, [&](int t, int) {
		const uint32_t r0 = (uint32_t)((uint64_t)fq->n * (uint64_t)t / (uint64_t)T), r1 = (uint32_t)((uint64_t)fq->n * (uint64_t)(t + 1) / (uint64_t)T);
		for (uint32_t r = r0; r < r1; ++r)
			for (int k = 0; k < 4; ++k)
				if (dest[r] >> k & 1) slice_bytes[(size_t)t * 4 + k] += rec_len(r, k);
	});
	int fds[4] = { -1, -1, -1, -1 };
	off_t start[4] = { 0, 0, 0, 0 };
	size_t total[4] = { 0, 0, 0, 0 };
	int rc = MNC_OK;
	for (int k = 0; k < 4 && rc == MNC_OK; ++k) {
		for (int t = 0; t < T; ++t) total[k] += slice_bytes[(size_t)t * 4 + k];
		if (total[k] == 0) continue;
		fds[k] = open(paths[k], O_WRONLY | O_CREAT, 0666);
		struct stat sb;
		if (fds[k] < 0 || fstat(fds[k], &sb) != 0) { set_error("cannot open %s: %s", paths[k], strerror(errno)); rc = MNC_ERR_IO; break; }
		start[k] = sb.st_size;                                // append: the new records follow what the file holds
	}
	int fail[4] = { 0, 0, 0, 0 };
	if (rc == MNC_OK) {
		team().run(T, [&](int t, int) {
			const uint32_t r0 = (uint32_t)((uint64_t)fq->n * (uint64_t)t / (uint64_t)T), r1 = (uint32_t)((uint64_t)fq->n * (uint64_t)(t + 1) / (uint64_t)T);
			off_t pos[4];
			std::string buf[4];
			for (int k = 0; k < 4; ++k) {
				pos[k] = start[k];
				for (int u = 0; u < t; ++u) pos[k] += (off_t)slice_bytes[(size_t)u * 4 + k];
			}
			auto flush = [&](int k) {
				std::string &b = buf[k];
				size_t done = 0;
				while (done < b.size()) {
					const ssize_t w = pwrite(fds[k], b.data() + done, b.size() - done, pos[k] + (off_t)done);
					if (w < 0 && errno == EINTR) continue;
					if (w <= 0) {
						__atomic_store_n(&fail[k], errno ? errno : EIO, __ATOMIC_RELAXED);
						break;
					}
					done += (size_t)w;
				}
				pos[k] += (off_t)b.size();
				b.clear();
			};
			for (uint32_t r = r0; r < r1; ++r) {
				const uint8_t d = dest[r];
				if (!d) continue;
				const char *tt = fq->titles.data() + fq->title_off[r];
				const size_t t_len = (size_t)(fq->title_off[r + 1] - fq->title_off[r]);
				const int64_t o = fq->offsets[r], l = fq->offsets[r + 1] - o;
				for (int k = 0; k < 4; ++k) {
					if (!(d >> k & 1)) continue;
					std::string &b = buf[k];
					b.push_back('@');
					if (k == 2 && label && label[r] >= 0) {           // seq_record.id = tax_unit
						const char *id = labels[label[r]];
						const size_t idl = strlen(id);
						if (t_len == 0) b.append(id, idl);
						else if (fq->id_len[r] == idl && memcmp(tt + fq->id_off[r], id, idl) == 0) b.append(tt, t_len);
						else { b.append(id, idl); b.push_back(' '); b.append(tt, t_len); }
					} else b.append(tt, t_len);
					b.push_back('\n');
					b.append(from_text ? text + fq->seq_off[r] : (const char*)fq->bases.p + o, (size_t)l);
					b.append("\n+\n", 3);
					b.append(from_text ? text + fq->qual_off[r] : (const char*)fq->quals.p + o, (size_t)l);
					b.push_back('\n');
					if (b.size() > (2u << 20)) flush(k);
				}
			}
			for (int k = 0; k < 4; ++k) if (!buf[k].empty()) flush(k);
		});
	}
	for (int k = 0; k < 4; ++k) {
		if (fds[k] < 0) continue;
		if (close(fds[k]) != 0 && !fail[k]) fail[k] = errno ? errno : EIO;
		if (rc == MNC_OK && fail[k]) { set_error("write to %s failed: %s", paths[k], strerror(fail[k])); rc = MNC_ERR_IO; }
	}
	return rc;
}

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
