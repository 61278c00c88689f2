// The FASTQ reader's pipeline as monica_amd/aligner.py drives it -- a thread parses and detaches batches, a second one
// routes them to the output files, a third asks for the qualities of the batch being routed -- on the CPU, for
// ThreadSanitizer: csrc/hostio.cpp with the three symbols it takes from the rest of the library defined here.  Batches
// share the reader's block by reference, blocks come from a process-wide cache, both passes run on teams (csrc/team.h).
// The routed files must hold what a plain loop over the records writes.
// g++ -O1 -g -std=c++17 -pthread -fsanitize=thread -Iinclude tests/hostio_tsan.cpp monica_amd/csrc/hostio.cpp -o hostio_tsan
#include "../include/monica_amd.h"
#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

namespace mnc { void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); } }
extern "C" void *mnc_host_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }
extern "C" void mnc_host_free(void *p) { free(p); }

struct Queue {
	std::mutex m; std::condition_variable cv; std::deque<mnc_fastq*> q; bool done = false;
	void put(mnc_fastq *b) { { std::lock_guard<std::mutex> g(m); q.push_back(b); } cv.notify_all(); }
	void finish() { { std::lock_guard<std::mutex> g(m); done = true; } cv.notify_all(); }
	mnc_fastq *get() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return done || !q.empty(); }); if (q.empty()) return nullptr; mnc_fastq *b = q.front(); q.pop_front(); return b; }
};

int main(int argc, char **argv)
{
	const std::string dir = argc > 1 ? argv[1] : "/tmp";
	const int n = argc > 2 ? atoi(argv[2]) : 30000;
	const std::string in = dir + "/in.fastq";
	const char *labels[2] = { "Escherichia_coli", "r7" };          // (r7: equal to one record's id -- the title is kept as it is)
	std::string want[3];
	{
		std::ofstream f(in, std::ios::binary);
		unsigned x = 99;
		for (int r = 0; r < n; ++r) {
			x = x * 1664525u + 1013904223u;
			const int len = 200 + (int)(x >> 20) % 1800;
			std::string seq((size_t)len, 'A'), qual((size_t)len, 'I');
			for (int i = 0; i < len; ++i) { x = x * 1664525u + 1013904223u; seq[(size_t)i] = "ACGT"[x >> 30]; qual[(size_t)i] = (char)(33 + (x >> 8) % 60); }
			std::ostringstream title;
			title << "r" << r << " ch=" << (r % 512) << " start_time=2026";
			f << '@' << title.str() << '\n' << seq << "\n+\n" << qual << '\n';
			const int k = r % 3;                                     // 0 unmapped, 1 ambiguous, 2 mapped under label r % 2
			std::string t = title.str();
			if (k == 2) { const std::string id = labels[r % 2]; if (t.substr(0, t.find(' ')) != id) t = id + " " + t; }
			want[k] += "@" + t + "\n" + seq + "\n+\n" + qual + "\n";
		}
	}
	const std::string out[3] = { dir + "/unmapped.fastq", dir + "/ambiguous.fastq", dir + "/mapped.fastq" };
	for (auto &p : out) remove(p.c_str());
	const char *paths[4] = { out[0].c_str(), out[1].c_str(), out[2].c_str(), nullptr };
	mnc_fastq *reader = nullptr;
	if (mnc_fastq_open(in.c_str(), &reader) != MNC_OK) return 2;
	Queue parsed;
	std::atomic<mnc_fastq*> routing{nullptr};
	std::atomic<bool> stop{false};
	std::atomic<int> fail{0};
	std::thread parser([&] {
		unsigned x = 7;
		for (;;) {
			x = x * 1664525u + 1013904223u;
			uint32_t got = 0;
			if (mnc_fastq_next(reader, 500 + (x >> 20) % 3000, 1u << 26, &got) != MNC_OK) { fail = 1; break; }
			if (!got) break;
			mnc_fastq *b = nullptr;
			if (mnc_fastq_detach_batch(reader, &b) != MNC_OK) { fail = 2; break; }
			parsed.put(b);
		}
		parsed.finish();
	});
	std::thread peeker([&] {                                           // the qualities of the batch in flight, as mnc_fastq_quals copies them out on first use
		while (!stop.load()) {
			mnc_fastq *b = routing.exchange(nullptr);
			if (b) { const uint8_t *q = mnc_fastq_quals(b); if (!q) fail = 3; routing.store(b); std::this_thread::yield(); }
		}
	});
	std::thread router([&] {
		int base = 0;
		while (mnc_fastq *b = parsed.get()) {
			const int64_t *off = mnc_fastq_offsets(b);
			uint32_t nb = 0;
			while (true) { const char *t; uint32_t l, il; if (mnc_fastq_title(b, nb, &t, &l, &il) != MNC_OK) break; ++nb; }
			(void)off;
			std::vector<uint8_t> dest(nb);
			std::vector<int32_t> label(nb);
			for (uint32_t r = 0; r < nb; ++r) { const int g = base + (int)r; dest[r] = (uint8_t)(1 << (g % 3)); label[r] = g % 3 == 2 ? g % 2 : -1; }
			routing.store(b);
			if (mnc_fastq_route(b, dest.data(), label.data(), labels, 2, paths) != MNC_OK) fail = 4;
			while (routing.exchange(nullptr) != b) std::this_thread::yield();   // (the peeker may hold it: take it back before it is freed)
			mnc_fastq_close(b);
			base += (int)nb;
		}
		if (base != n) fail = 5;
	});
	parser.join();
	router.join();
	stop = true;
	peeker.join();
	mnc_fastq_close(reader);
	for (int k = 0; k < 3; ++k) {
		std::ifstream f(out[k], std::ios::binary);
		std::stringstream ss; ss << f.rdbuf();
		if (ss.str() != want[k]) { fprintf(stderr, "%s differs: %zu bytes against %zu\n", out[k].c_str(), ss.str().size(), want[k].size()); fail = 6; }
	}
	if (fail.load()) { printf("failed %d\n", fail.load()); return 1; }
	puts("ok");
	return 0;
}
