// Host test of csrc/team.h (the parse and routing passes' helper threads): every index of every pass is visited exactly
// once whatever the width, passes of changing widths follow one another on one team, teams of several calling threads run
// side by side, a pass inside a pass uses the helper's own team, and a thread's team ends with the thread.
// g++ -O2 -std=c++17 -pthread [-fsanitize=thread] -o team_host tests/team_host.cpp && ./team_host
#include "../monica_amd/csrc/team.h"
#include <atomic>
#include <cstdio>
#include <numeric>

using mnc::team;

static bool one_thread(int seed)
{
	bool ok = true;
	std::vector<int> hits(100003);
	unsigned x = 12345u + (unsigned)seed;
	for (int round = 0; round < 300 && ok; ++round) {
		x = x * 1664525u + 1013904223u;
		const int T = 1 + (int)(x >> 24) % 16;
		const int64_t n = (int64_t)(x >> 8) % (int64_t)hits.size();
		std::fill(hits.begin(), hits.end(), 0);
		team().slices(T, n, [&](int64_t i) { ++hits[(size_t)i]; });          // disjoint slices: no two threads share an index
		for (int64_t i = 0; i < (int64_t)hits.size(); ++i) ok = ok && hits[(size_t)i] == (i < n ? 1 : 0);
		std::atomic<int> calls{0}, width_seen{0};
		team().run(T, [&](int t, int nt) { calls.fetch_add(1); if (t == nt - 1) width_seen.store(nt); });
		ok = ok && calls.load() == T && width_seen.load() == T;
	}
	// a pass inside a pass: the helper's own team
	std::atomic<long long> sum{0};
	team().run(4, [&](int t, int) {
		team().slices(3, 1000, [&](int64_t i) { sum.fetch_add(i + t); });
	});
	long long want = 0;
	for (int t = 0; t < 4; ++t) for (int i = 0; i < 1000; ++i) want += i + t;
	return ok && sum.load() == want;
}

int main()
{
	std::atomic<int> good{0};
	std::vector<std::thread> callers;
	for (int c = 0; c < 4; ++c) callers.emplace_back([&, c] { if (one_thread(c)) good.fetch_add(1); });   // (their teams end with them)
	for (auto &t : callers) t.join();
	const bool main_ok = one_thread(99);
	if (good.load() == 4 && main_ok) { puts("ok"); return 0; }
	printf("failed: %d of 4 threads, main %d\n", good.load(), (int)main_ok);
	return 1;
}
