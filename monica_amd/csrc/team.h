// Host threads of one parse or routing pass.  A calling thread owns a team of helpers that SLEEP between passes:
// the aligner's pipeline has a parser thread and a routing thread side by side with the thread that launches
// kernels, and an OpenMP runtime keeps the idle helpers of each spinning -- on the GPU box (a 16-core share of a
// 256-thread host) the spinning was 0.7 of the 2.0 core-seconds a gigabyte of FASTQ took, and the share's quota
// was what the run waited for (profiles/r05y_files_cpu.txt).
#pragma once

#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace mnc {

class Team {
	std::mutex m;
	std::condition_variable go, done;
	std::vector<std::thread> helpers;
	const std::function<void(int, int)> *job = nullptr;
	int width = 0, pending = 0;
	uint64_t gen = 0;
	bool quit = false;
	bool busy = false;                                         // the owner is inside run(): a pass started from within a pass runs on the caller alone

	void loop(int id, uint64_t seen)
	{
		std::unique_lock<std::mutex> lk(m);
		for (;;) {
			go.wait(lk, [&] { return quit || gen != seen; });
			if (quit) return;
			seen = gen;
			if (id >= width) continue;                          // a narrower pass than the team
			const std::function<void(int, int)> *f = job;
			const int w = width;
			lk.unlock();
			(*f)(id, w);
			lk.lock();
			if (--pending == 0) done.notify_one();
		}
	}

public:
	Team() = default;
	Team(const Team &) = delete;
	Team &operator=(const Team &) = delete;
	~Team()
	{
		{
			std::lock_guard<std::mutex> lk(m);
			quit = true;
		}
		go.notify_all();
		for (auto &t : helpers) t.join();
	}
	// fn(t, T) for t = 0 .. T - 1, t = 0 on the calling thread; returns when all are done.  fn must not throw.
	void run(int T, const std::function<void(int, int)> &fn)
	{
		if (T <= 1 || busy) {                                    // (busy: only the owner's own share of a pass can get here -- its helpers have teams of their own)
			for (int t = 0; t < (T < 1 ? 1 : T); ++t) fn(t, T < 1 ? 1 : T);
			return;
		}
		struct Busy { bool &b; Busy(bool &b_) : b(b_) { b = true; } ~Busy() { b = false; } } in_pass(busy);
		{
			std::lock_guard<std::mutex> lk(m);
			while ((int)helpers.size() < T - 1) {
				const int id = (int)helpers.size() + 1;
				helpers.emplace_back(&Team::loop, this, id, gen);
			}
			job = &fn, width = T, pending = T - 1, ++gen;
		}
		go.notify_all();
		fn(0, T);
		std::unique_lock<std::mutex> lk(m);
		done.wait(lk, [&] { return pending == 0; });
	}
	// i = 0 .. n - 1 in T contiguous slices
	template <class F> void slices(int T, int64_t n, F &&body)
	{
		run(T, [&](int t, int nt) {
			const int64_t a = n * t / nt, b = n * (t + 1) / nt;
			for (int64_t i = a; i < b; ++i) body(i);
		});
	}
};

// the calling thread's team (its helpers end with the thread)
inline Team &team()
{
	thread_local Team t;
	return t;
}

} // namespace mnc
