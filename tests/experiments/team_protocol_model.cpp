// team_protocol_model.cpp — a host-thread model of the control words of the beam kernel's team form
// (pg_embedding_amd/csrc/device_search.h: TeamCtl, team_help, the walk loop of hnsw_search_kernel_beam<…, TEAM = true>).
//
// Test infrastructure, not product: every wave is a thread, every LDS word an atomic, every stretch of work a random
// spin.  It answers two questions the device suites cannot, because they need a schedule that almost never happens:
//
//   1. A wave that walks a SECOND query while siblings still help it: can it read a package that a helper scored
//      against the FIRST query?  (state goes 1 -> 0 -> 1; a helper in the middle of a step across the whole gap never
//      sees the 0.)   Model flag `clear`: the walking wave clears its helper bits before it opens a walk (what ships).
//   2. The slice-helper jobs of scripts/pending/slice_helpers_and_bulk_append.patch: can the walking wave wait for ever
//      for a helper it named?   Model flag `early`: the helper reads the job counter BEFORE its bit becomes visible
//      (the patch) instead of after (the version whose last device run never returned).
//
// usage: team_protocol_model <clear 0|1> <early 0|1> <slices 0|1> <walks> <seed> [hang limit in ms, default 20000]
// prints: stale=<packages of another query that were consumed> hangs=<jobs never completed> jobs=<posted> pk=<consumed>
// exit status 0 when both are zero.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

namespace {

constexpr int HELPERS = 7;              // waves 1..7 of a block help wave 0
constexpr uint32_t PASS = 8;            // rows of one scoring pass = one slice

struct Ctl
{
	std::atomic<uint32_t> state{0};     // 0 = between walks, 1 = walking, 2 = never again
	std::atomic<uint32_t> helpers{0};
	std::atomic<uint32_t> job{0};       // seq << 18 | view << 15 | mask << 7 | rows
	std::atomic<uint32_t> done{0};
};

struct Region                           // what a helper offers: one package, tagged with the query it was scored against
{
	std::atomic<uint64_t> pkg{0};       // 0 = none, else (query + 1) << 1 | 1
};

Ctl ctl;
Region region[HELPERS + 1];
std::atomic<int> loaded_query{-1};      // the query in the walking wave's region
std::atomic<uint32_t> slice_mark[64];   // the walking wave's sum array: which job scored row r
std::atomic<uint64_t> stale{0}, hangs{0}, jobs{0}, consumed{0};
bool opt_clear, opt_early, opt_slices;
int opt_limit_ms = 20000;           // a job not completed after this long counts as a hang

struct Rng
{
	uint64_t s;
	uint32_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t) (s >> 11); }
	void spin(uint32_t max)             // a stretch of work of random length
	{
		const uint32_t n = next() % (max + 1);
		for (volatile uint32_t i = 0; i < n; i++) {}
		if ((next() & 1023u) == 0) std::this_thread::yield();
	}
};

void helper(int wib, uint64_t seed)
{
	Rng r{seed * 0x9E3779B97F4A7C15ull + (uint64_t) wib};
	const uint32_t bit = 1u << wib;
	for (;;)
	{
		const uint32_t st = ctl.state.load();
		if (st == 2) return;
		if (st != 1) { r.spin(20); std::this_thread::yield(); continue; }
		// my region becomes that walk's package store
		region[wib].pkg.store(0);
		r.spin(60);                                             // (clearing the memo, the package headers ...)
		uint32_t last_job = 0;
		if (opt_early) last_job = ctl.job.load() >> 18;
		ctl.helpers.fetch_or(bit);
		r.spin(200);                                            // (query norm for the cosine metric)
		if (!opt_early) last_job = ctl.job.load() >> 18;
		while (ctl.state.load() == 1)
		{
			const uint32_t hm = ctl.helpers.load();
			if (!(hm & bit)) break;
			const uint32_t myrank = (uint32_t) __builtin_popcount(hm & (bit - 1u));
			if (opt_slices)
			{
				const uint32_t job = ctl.job.load();
				if ((job >> 18) != last_job)
				{
					last_job = job >> 18;
					const uint32_t jm = (job >> 7) & 0xFFu, jn = job & 0x7Fu;
					if (jm & bit)
					{
						const uint32_t lo = PASS + PASS * (uint32_t) __builtin_popcount(jm & (bit - 1u));
						const uint32_t cnt = jn - lo < PASS ? jn - lo : PASS;
						r.spin(300);                            // (one scoring pass)
						for (uint32_t i = 0; i < cnt; i++) slice_mark[lo + i].store(job >> 18);
						ctl.done.fetch_add(1);
					}
					continue;
				}
				if (myrank >= 5) { r.spin(10); std::this_thread::yield(); continue; }      // a slice helper does not speculate
			}
			// one speculation step: link list + scoring against the query that is in the walking wave's region NOW
			const int q = loaded_query.load();
			r.spin(3000);
			region[wib].pkg.store(((uint64_t) (q + 1) << 1) | 1u);
		}
		ctl.helpers.fetch_and(~bit);
	}
}

void walker(int walks, uint64_t seed)
{
	Rng r{seed};
	uint32_t jobseq = 0;
	for (int w = 0; w < walks; w++)
	{
		loaded_query.store(w);
		r.spin(1500);                                           // (query load, entry point scoring)
		if (opt_clear) ctl.helpers.store(0);
		ctl.state.store(1);
		const int hops = 20 + (int) (r.next() % 40);
		for (int h = 0; h < hops; h++)
		{
			const uint32_t hm = ctl.helpers.load();
			r.spin(400);                                        // (pop, link list)
			for (int x = 1; x <= HELPERS; x++)
				if (hm & (1u << x))
				{
					const uint64_t p = region[x].pkg.load();
					if (p & 1u)
					{
						consumed++;
						if ((int) (p >> 1) - 1 != w) stale++;
					}
				}
			const uint32_t nscore = 1 + r.next() % 32;
			if (opt_slices && hm && nscore > PASS)
			{
				uint32_t m = hm;
				for (uint32_t i = 0; i < 5 && m; i++) m &= m - 1;
				const uint32_t want = (nscore - 1) / PASS;
				uint32_t jm = 0;
				for (uint32_t i = 0; i < want && m; i++) { jm |= m & (0u - m); m &= m - 1; }
				const uint32_t nsl = (uint32_t) __builtin_popcount(jm);
				if (nsl)
				{
					ctl.done.store(0);
					jobseq = (jobseq + 1) & 0x3FFFu;
					ctl.job.store((jobseq << 18) | ((uint32_t) __builtin_ctz(hm) << 15) | (jm << 7) | nscore);
					jobs++;
					r.spin(300);                                // (my own pass)
					const auto t0 = std::chrono::steady_clock::now();
					bool hung = false;
					while (ctl.done.load() != nsl)                 // (a true hang is for ever: the limit only has to outlast a busy machine)
					{
						if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(opt_limit_ms)) { hung = true; break; }
						std::this_thread::yield();
					}
					if (hung)
					{
						hangs++;
						ctl.state.store(2);                     // the device would sit here for ever
						return;
					}
					const uint32_t covered = PASS + PASS * nsl < nscore ? PASS + PASS * nsl : nscore;
					for (uint32_t i = PASS; i < covered; i++)
						if (slice_mark[i].load() != jobseq) stale++;      // a row nobody scored for THIS job
				}
			}
			r.spin(300);                                        // (accept loop)
		}
		ctl.state.store(0);
		r.spin(1200);                                           // (emit: select, order, label look-up, clean-up)
	}
	ctl.state.store(2);
}

}  // namespace

int main(int argc, char **argv)
{
	if (argc < 6) { fprintf(stderr, "usage: %s clear early slices walks seed\n", argv[0]); return 2; }
	opt_clear = atoi(argv[1]) != 0; opt_early = atoi(argv[2]) != 0; opt_slices = atoi(argv[3]) != 0;
	const int walks = atoi(argv[4]);
	const uint64_t seed = strtoull(argv[5], nullptr, 10) | 1u;
	if (argc > 6) opt_limit_ms = atoi(argv[6]);
	for (auto &m : slice_mark) m.store(0);
	std::vector<std::thread> th;
	for (int h = 1; h <= HELPERS; h++) th.emplace_back(helper, h, seed);
	std::thread w(walker, walks, seed);
	w.join();
	for (auto &t : th) t.join();
	printf("clear=%d early=%d slices=%d walks=%d stale=%llu hangs=%llu jobs=%llu pk=%llu\n", (int) opt_clear, (int) opt_early,
		   (int) opt_slices, walks, (unsigned long long) stale.load(), (unsigned long long) hangs.load(),
		   (unsigned long long) jobs.load(), (unsigned long long) consumed.load());
	return stale.load() || hangs.load() ? 1 : 0;
}
