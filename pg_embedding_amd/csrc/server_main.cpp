// server_main.cpp — hnsw_gpu_server: the GPU-owning process behind libembedding_gpuc.so
// (include/hnsw_gpu_server.h has the why and the wire format).
//
//   readers      epoll threads: parse requests; answer the trivial ones; SEARCH goes to the batch
//                queue, everything that changes a mirror goes to the control thread
//   dispatchers  (default 2) each keep up to --lanes (default 3) launches in flight.  A free lane
//                takes every queued SEARCH for one (mirror, ef) — up to --max-batch — packs the
//                queries into the lane's pinned host buffer and launches the fused search kernel on
//                the lane's own context (own visited-set workspace, own HIP stream) with the result
//                arrays and one completion flag per query IN that pinned buffer
//                (hnsw_gpu_search_batch_ctx_flags): nothing is copied, nothing is waited for.  The
//                dispatcher polls the flags and answers each backend the moment ITS walk has ended —
//                a walk is ~160 dependent hops and the slowest of a batch takes 2-3x the mean.
//                Requests gather while the lanes are busy: no artificial delay unless --linger-us
//                is given.  --lanes 0 = one blocking launch at a time per dispatcher.
//   control      one thread: UPLOAD / UPDATE / BIND / LINK / EXPORT / DROP / SET_DELETED / SETGEN / DIST.
//                Mirror changes wait for the mirror's searches in flight and keep new ones out
//                (counting gate, writer first).
//
// All device work goes through the C API of libhnsw_gpu.so (hnsw_gpu.h); this file has no HIP in
// it and no arithmetic.  No device, no service: exit status 3.
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <csignal>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include <fcntl.h>
#include <pthread.h>
#include <sys/epoll.h>
#include <sys/eventfd.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <sys/un.h>
#include <linux/futex.h>

#include "hgs_io.h"
#include "hnsw_gpu.h"

namespace {

// Timed waits: condition_variable::wait_for — except in the ThreadSanitizer build of the CPU tier (-DHGS_TSAN,
// tests/server_util.py), where they go through the wall clock (pthread_cond_timedwait): gcc 11's ThreadSanitizer does not know
// pthread_cond_clockwait, believes the mutex stays held across the wait and reports every queue access behind it.
#ifdef HGS_TSAN
#define HGS_TIMED_WAIT(cv, lk, d, ...) (cv).wait_until((lk), std::chrono::system_clock::now() + (d), __VA_ARGS__)
#else
#define HGS_TIMED_WAIT(cv, lk, d, ...) (cv).wait_for((lk), (d), __VA_ARGS__)
#endif

// ----------------------------------------------------------------------------- options / globals
struct Options
{
	std::string path;
	int device = 0;
	int readers = 4;
	int dispatchers = 2;
	size_t max_batch = 16384;
	int lanes = 3;                 // launches in flight per dispatcher (0 = one blocking launch at a time)
	long linger_us = 0;
	size_t min_batch = 1;          // with --linger-us: wait while fewer requests than this are queued
	int backlog = 1024;
	bool verbose = false;
	int ready_fd = -1;
	int walkers = -1;              // walking waves per block of a search launch: -1 = by load (below), 0 = the library's default, 1..8 fixed
	int stream = 0;                // 1 = searches go through ONE resident launch per (mirror, ef) fed through a pinned ring (stream_manager)
	size_t ring = 4096;            // slots of that ring
	std::vector<std::string> shard_peers;   // sockets of the servers that hold the OTHER row shards of every mirror: this server is their front
	                               // (include/hnsw_gpu_server.h, HGS_OP_SHARD_*); needs --lanes 0 --stream 0
	int shm_pollers = 0;           // threads that poll the backends' mailboxes (HGS_OP_SHM); 0 (default) = mailboxes are refused: every
	                               // request on the socket.  Opt-in: measured, the mailboxes gain nothing on a host whose CPU time is capped
	                               // (the pollers spin), profiles/r4aj_mailboxes.txt
} g_opt;

std::atomic<bool> g_stop{false};
int g_wake_fd = -1;                // eventfd: wakes every epoll loop at shutdown
std::chrono::steady_clock::time_point g_t0;

// Walks in flight over ALL lanes (queries launched and not yet answered) and the 8-wave team blocks the device holds: a lone small
// launch gives every walk a block of its own (seven helpers: the latency shape), but the lanes' launches share ONE device — with W
// walks in flight, ceil(W / blocks) waves per block must walk or the helpers of some walks keep other walks off the device
// (include/hnsw_gpu.h, hnsw_gpu_ctx_set_walkers; profiles/r4d_server_sweep.txt).
std::atomic<long> g_walks_in_flight{0};
int g_device_blocks = 256;

struct Counters
{
	std::atomic<uint64_t> connections{0}, connections_now{0}, searches{0}, batches{0}, max_batch{0},
		search_errors{0}, uploads{0}, upload_bytes{0}, updates{0}, binds{0}, evictions{0}, batch_ns{0}, kernel_ns{0},
		queue_ns{0}, walk_ns{0}, answer_ns{0}, shm_searches{0};
} g_cnt;

uint64_t now_ns()
{
	return (uint64_t) std::chrono::duration_cast<std::chrono::nanoseconds>(
		std::chrono::steady_clock::now().time_since_epoch()).count();
}

void logf(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	fprintf(stderr, "hnsw_gpu_server: ");
	vfprintf(stderr, fmt, ap);
	fputc('\n', stderr);
	va_end(ap);
}
#define VLOG(...) do { if (g_opt.verbose) logf(__VA_ARGS__); } while (0)

// ----------------------------------------------------------------------------- mirrors
static std::atomic<uint64_t> g_vseq{0};          // content versions: never repeat within a server's life
// "Absent" has a version too: the value LOOKUP reports for a key without a mirror, new at every removal of ANY mirror
// (DROP — also of a key that had none —, eviction, an upload that dropped its predecessor and then failed).  Without
// it "no mirror" at LOOKUP and "no mirror" at UPLOAD looked the same (0 == 0) across a VACUUM that flipped flags in
// place and dropped the mirror in between — and the guarded upload of a snapshot walked before the VACUUM was taken,
// with its stale flags and an unchanged generation.  One server-wide epoch is conservative (any removal makes the
// guarded uploads in flight for absent keys walk again), never wrong.
static std::atomic<uint64_t> g_absent_ver{0};
static void note_removed() { g_absent_ver.store(++g_vseq); }

struct Entry
{
	uint64_t key = 0;
	std::atomic<uint64_t> gen{0};
	HnswMetadata meta;
	hnsw_gpu_index *ix = nullptr;
	// searches: readers (several launches of one dispatcher thread may hold it at once, so this is a
	// counting gate, not a pthread rwlock); BIND/UPDATE/LINK/...: one writer, and a waiting writer
	// keeps new searches out
	std::mutex gmu;
	std::condition_variable gcv;
	int readers = 0, writers_waiting = 0;
	bool writing = false;
	std::mutex cmu;
	std::vector<hnsw_gpu_ctx *> ctx;             // one per dispatcher lane, made on first use
	std::atomic<uint64_t> last_used{0};
	std::atomic<size_t> count{0};
	// Content version: a server-wide sequence number taken at creation and at every change (BIND, UPDATE, LINK,
	// vacuum flag, rename).  LOOKUP reports it and an UPLOAD built from a walk that started at that LOOKUP is only
	// accepted while it still holds (0 = no mirror): a snapshot can never silently replace a mirror that an
	// inserter or another uploader has changed in between, whatever the generations are called.
	std::atomic<uint64_t> version{0};

	Entry() { memset(&meta, 0, sizeof(meta)); }
	~Entry()
	{
		for (hnsw_gpu_ctx *c : ctx) if (c) hnsw_gpu_ctx_destroy(c);
		if (ix) hnsw_gpu_index_destroy(ix);
	}
	bool try_read()
	{
		std::lock_guard<std::mutex> lk(gmu);
		if (writing || writers_waiting) return false;
		readers++;
		return true;
	}
	void begin_read()
	{
		std::unique_lock<std::mutex> lk(gmu);
		gcv.wait(lk, [this] { return !writing && !writers_waiting; });
		readers++;
	}
	void end_read()
	{
		std::lock_guard<std::mutex> lk(gmu);
		if (--readers == 0) gcv.notify_all();
	}
	bool writer_wants()            // somebody is changing, or waiting to change, this mirror (a stream session gives way)
	{
		std::lock_guard<std::mutex> lk(gmu);
		return writing || writers_waiting > 0;
	}
	void begin_write()
	{
		std::unique_lock<std::mutex> lk(gmu);
		writers_waiting++;
		gcv.wait(lk, [this] { return readers == 0 && !writing; });
		writers_waiting--;
		writing = true;
	}
	void end_write()
	{
		std::lock_guard<std::mutex> lk(gmu);
		writing = false;
		gcv.notify_all();
	}
	hnsw_gpu_ctx *context(int d)
	{
		std::lock_guard<std::mutex> lk(cmu);
		if (ctx.size() <= (size_t) d) ctx.resize((size_t) d + 1, nullptr);
		if (!ctx[d] && hnsw_gpu_ctx_create(ix, &ctx[d]) != HNSW_GPU_OK) ctx[d] = nullptr;
		return ctx[d];
	}
};
using EntryP = std::shared_ptr<Entry>;

std::mutex g_map_mu;
std::unordered_map<uint64_t, EntryP> g_map;

EntryP find_entry(uint64_t key)
{
	std::lock_guard<std::mutex> lk(g_map_mu);
	auto it = g_map.find(key);
	return it == g_map.end() ? EntryP() : it->second;
}

// Drop the least recently used mirror nobody else holds (not `keep`).  True if one was dropped.
bool evict_one(uint64_t keep)
{
	EntryP victim;
	{
		std::lock_guard<std::mutex> lk(g_map_mu);
		uint64_t best = ~0ull;
		for (auto &kv : g_map)
			if (kv.first != keep && kv.second.use_count() == 1 && kv.second->last_used.load() < best)
			{
				best = kv.second->last_used.load();
				victim = kv.second;
			}
		if (victim) g_map.erase(victim->key);
	}
	if (!victim) return false;
	note_removed();
	logf("evicting mirror %llx (%zu elements) to make room", (unsigned long long) victim->key, victim->count.load());
	g_cnt.evictions++;
	return true;        // freed when `victim` goes out of scope here
}

// ----------------------------------------------------------------------------- connections
const int MAX_INFLIGHT_PER_CONN = 64;

struct Conn
{
	int fd = -1;
	std::vector<char> in;
	std::deque<int> fds;          // descriptors received and not yet claimed by a request
	std::mutex wmu;
	std::atomic<bool> closed{false};
	std::atomic<int> inflight{0};  // SEARCH requests not answered yet (a backend has one; the cap stops a flood)
	// the backend's mailbox (HGS_OP_SHM, include/hnsw_gpu_server.h): mapped here; capacities from the file's size
	hgs_shm *shm = nullptr;
	size_t shm_bytes = 0;
	uint32_t shm_qcap = 0, shm_rcap = 0;
	std::atomic<bool> shm_busy{false};   // a posted request has been taken and not answered yet (server-side word: the backend cannot touch it)
	// a FRONT's exchange buffer mapped for this connection (HGS_OP_SHARD_ATTACH; the control thread is the only user)
	void *shard_buf = nullptr;
	size_t shard_bytes = 0;
	~Conn()
	{
		for (int f : fds) close(f);
		if (fd >= 0) close(fd);
		if (shm) munmap(shm, shm_bytes);
		if (shard_buf) (void) hnsw_gpu_shared_close(g_opt.device, shard_buf);
	}
	// the answer of a request that was posted through the mailbox
	void respond_shm(const hgs_hdr &h, const void *p1, size_t l1, const void *p2, size_t l2)
	{
		char *base = reinterpret_cast<char *>(shm) + HGS_SHM_DATA + (size_t) shm_qcap * 4u;
		hgs_hdr out = h;
		if (l1 > (size_t) shm_rcap * 8u || l2 > (size_t) shm_rcap * 4u) { out.status = (int16_t) HGS_ERR_PROTOCOL; out.len = 0; l1 = l2 = 0; }
		if (l1) memcpy(base, p1, l1);
		if (l2) memcpy(base + (size_t) shm_rcap * 8u, p2, l2);
		memcpy(&shm->resp, &out, sizeof(out));
		__atomic_store_n(&shm->state, (uint32_t) HGS_SHM_DONE, __ATOMIC_SEQ_CST);
		shm_busy.store(false, std::memory_order_release);       // (after DONE: a poller that saw "not busy" while the word still said POSTED would take the request twice)
		// the backend says when it sleeps (it stores `sleeping` before it looks at `state` a last time inside futex_wait, we store `state`
		// before we look at `sleeping`): either it sees DONE and does not sleep, or we see it sleeping and wake it
		if (__atomic_load_n(&shm->sleeping, __ATOMIC_SEQ_CST))
			(void) syscall(SYS_futex, &shm->state, FUTEX_WAKE, 1, nullptr, nullptr, 0);
	}
	void respond(const hgs_hdr &req, int status, uint64_t a0 = 0, uint64_t a1 = 0, const void *p1 = nullptr,
				 size_t l1 = 0, const void *p2 = nullptr, size_t l2 = 0, uint64_t gen = 0)
	{
		if (req.op == HGS_OP_SEARCH) inflight--;
		if (closed.load()) return;
		hgs_hdr h;
		memset(&h, 0, sizeof(h));
		h.magic = HGS_MAGIC; h.op = req.op; h.status = (int16_t) status;
		h.len = (uint32_t) ((p1 ? l1 : 0) + (p2 ? l2 : 0));
		h.key = req.key; h.gen = gen ? gen : req.gen; h.a0 = a0; h.a1 = a1;
		if (req.magic == HGS_SHM_MAGIC) { respond_shm(h, p1, p1 ? l1 : 0, p2, p2 ? l2 : 0); return; }
		std::lock_guard<std::mutex> lk(wmu);
		// a backend that does not read its answer may hold a dispatcher for 2 s, once: then it is cut off
		if (hgs::send_msg(fd, &h, p1, l1, p2, l2, -1, 2000) != 0) { closed.store(true); shutdown(fd, SHUT_RDWR); }
	}
};
using ConnP = std::shared_ptr<Conn>;

// ----------------------------------------------------------------------------- queues
struct SReq            // one backend's hnsw_search
{
	ConnP c;
	EntryP e;
	hgs_hdr h;
	std::vector<float> q;
	uint64_t t_in = 0;             // arrival (parsed by a reader), for hgs_stats.queue_ns
};
std::mutex g_q_mu;
std::condition_variable g_q_cv;
std::deque<SReq> g_q;
std::atomic<long> g_q_waiting{0};          // stream mode: requests in g_q (the readers' "is anything queued" without the lock)

// Stream mode: control work that needs the DEVICE — a mirror destroyed (hipFree), an upload's build or link kernels, a staging
// re-allocation, a hipDeviceSynchronize — can wait for ever behind a resident launch, which holds every block slot and never ends by
// itself while searches keep coming (the session only ever yielded to a writer on ITS OWN mirror).  The control thread counts itself
// in here around such work; the stream manager closes the open session while the count is up and opens none until it is down.
std::atomic<int> g_device_wanted{0};
struct DeviceWanted
{
	DeviceWanted() { g_device_wanted.fetch_add(1); }
	~DeviceWanted() { g_device_wanted.fetch_sub(1); }
};

struct CReq            // a control request
{
	ConnP c;
	hgs_hdr h;
	std::vector<char> payload;
	int fd = -1;
};
std::mutex g_c_mu;
std::condition_variable g_c_cv;
std::deque<CReq> g_c;

// ----------------------------------------------------------------------------- dispatchers
struct Pinned
{
	void *p = nullptr; size_t bytes = 0;
	~Pinned() { if (p) hnsw_gpu_host_free(p); }
	bool reserve(size_t want)
	{
		if (want <= bytes) return true;
		if (p) hnsw_gpu_host_free(p);
		p = nullptr; bytes = 0;
		size_t nb = (size_t) 1 << 20;
		while (nb < want) nb *= 2;
		p = hnsw_gpu_host_alloc(nb);
		if (!p) return false;
		bytes = nb;
		return true;
	}
};

// ---- the front of a row-sharded index (--shard-peers; include/hnsw_gpu_server.h, HGS_OP_SHARD_*) ----------------------------------
// Per dispatcher thread: a connection to every peer and ONE exchange buffer (device memory of this server, shared with the peers
// through its IPC handle): [list][nq][ef] labels, then [list][nq][ef] distances, list 0 = this server's own shard.
struct ShardFront
{
	std::vector<int> fds;                 // one per peer (-1 = not connected)
	void *xbuf = nullptr; size_t xbytes = 0;
	hnsw_gpu_ipc_handle handle;
	std::vector<bool> attached;           // the peer has mapped the CURRENT buffer
	~ShardFront()
	{
		for (int f : fds) if (f >= 0) close(f);
		if (xbuf) (void) hnsw_gpu_shared_free(g_opt.device, xbuf);
	}
	// one request / response on a peer's connection; false = the connection is gone (closed here)
	bool call(size_t k, hgs_hdr &h, const void *payload, size_t len, int timeout_ms)
	{
		h.magic = HGS_MAGIC; h.status = 0; h.len = (uint32_t) len;
		hgs_hdr resp;
		int got = -1;
		if (hgs::send_msg(fds[k], &h, payload, len, nullptr, 0, -1, timeout_ms) != 0 ||
			hgs::recv_exact(fds[k], &resp, sizeof(resp), &got, timeout_ms) != 0 || resp.magic != HGS_MAGIC || resp.op != h.op || resp.len > 64)
		{
			if (got >= 0) close(got);
			close(fds[k]); fds[k] = -1; attached[k] = false;
			return false;
		}
		char skip[64];
		if (resp.len && hgs::recv_exact(fds[k], skip, resp.len, &got, timeout_ms) != 0) { close(fds[k]); fds[k] = -1; attached[k] = false; return false; }
		if (got >= 0) close(got);
		h = resp;
		return true;
	}
	bool connect_peer(size_t k)
	{
		if (fds[k] >= 0) return true;
		const std::string &path = g_opt.shard_peers[k];
		struct sockaddr_un addr;
		memset(&addr, 0, sizeof(addr));
		addr.sun_family = AF_UNIX;
		if (path.size() >= sizeof(addr.sun_path)) return false;
		strncpy(addr.sun_path, path.c_str(), sizeof(addr.sun_path) - 1);
		const int fd = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
		if (fd < 0) return false;
		if (connect(fd, (struct sockaddr *) &addr, sizeof(addr)) != 0) { close(fd); return false; }
		fds[k] = fd;
		hgs_hdr h;
		memset(&h, 0, sizeof(h));
		h.op = HGS_OP_HELLO; h.a0 = HGS_VERSION;
		if (!call(k, h, nullptr, 0, 5000) || h.status != HGS_OK) { if (fds[k] >= 0) { close(fds[k]); fds[k] = -1; } return false; }
		return true;
	}
	// room for `lists` lists of nq x ef: (re)allocates and re-attaches every peer when it grows
	int ensure(size_t bytes)
	{
		if (fds.empty()) { fds.assign(g_opt.shard_peers.size(), -1); attached.assign(g_opt.shard_peers.size(), false); }
		if (bytes > xbytes)
		{
			if (xbuf) (void) hnsw_gpu_shared_free(g_opt.device, xbuf);
			xbuf = nullptr; xbytes = 0;
			size_t nb = (size_t) 4 << 20;
			while (nb < bytes) nb *= 2;
			const int rc = hnsw_gpu_shared_alloc(g_opt.device, nb, &xbuf, &handle);
			if (rc != HNSW_GPU_OK) { xbuf = nullptr; return rc; }
			xbytes = nb;
			attached.assign(attached.size(), false);
		}
		for (size_t k = 0; k < fds.size(); k++)
		{
			if (!connect_peer(k)) { fail_text = "cannot reach shard peer " + g_opt.shard_peers[k]; return HGS_ERR_IO; }
			if (attached[k]) continue;
			hgs_hdr h;
			memset(&h, 0, sizeof(h));
			h.op = HGS_OP_SHARD_ATTACH; h.a0 = xbytes;
			if (!call(k, h, &handle, sizeof(handle), 10000) || h.status != HGS_OK)
			{
				fail_text = "shard peer " + g_opt.shard_peers[k] + " did not map the exchange buffer";
				return HGS_ERR_IO;
			}
			attached[k] = true;
		}
		return HNSW_GPU_OK;
	}
	std::string fail_text;
};

// One batch on every shard + the merge.  Q (nq x dim), the merged labels / distances / counts live in `pin` (the kernels read and
// write pinned host memory directly); returns an HGS / HNSW_GPU status.
int run_batch_sharded(int d, Entry *e, size_t nq, size_t ef, const float *Q, label_t *L, dist_t *D, uint32_t *C, uint32_t *scratch_counts)
{
	static thread_local ShardFront sf;
	const size_t lists = g_opt.shard_peers.size() + 1, dim = e->meta.dim;
	const size_t lb = (lists * nq * ef * 8 + 255) & ~(size_t) 255, db = lists * nq * ef * 4;
	int rc = sf.ensure(lb + db);
	if (rc != HNSW_GPU_OK) { logf("sharded batch: %s", sf.fail_text.empty() ? hnsw_gpu_last_error() : sf.fail_text.c_str()); return rc; }
	char *xb = (char *) sf.xbuf;
	// the peers first (their searches run while ours does) ...
	for (size_t k = 0; k < sf.fds.size(); k++)
	{
		hgs_hdr h;
		memset(&h, 0, sizeof(h));
		h.magic = HGS_MAGIC; h.op = HGS_OP_SHARD_SEARCH; h.key = e->key; h.aux = (uint32_t) ef; h.a0 = nq; h.len = (uint32_t) (nq * dim * 4);
		h.a1 = (1 + k) * nq * ef * 8;                       // this peer's labels ...
		h.gen = lb + (1 + k) * nq * ef * 4;                 // ... and distances
		if (hgs::send_msg(sf.fds[k], &h, Q, nq * dim * 4, nullptr, 0, -1, 10000) != 0)
		{
			close(sf.fds[k]); sf.fds[k] = -1; sf.attached[k] = false;
			logf("sharded batch: lost shard peer %s", g_opt.shard_peers[k].c_str());
			// (the peers already asked answer into a buffer nobody merges; their connections are re-synchronised by the reads below)
			rc = HGS_ERR_IO;
		}
	}
	// ... then this server's own shard into list 0
	hnsw_gpu_ctx *ctx = e->context(d);
	if (rc == HNSW_GPU_OK && !ctx) rc = HNSW_GPU_ERR_HIP;
	if (rc == HNSW_GPU_OK)
		rc = hnsw_gpu_search_batch_ctx(ctx, Q, nq, ef, (label_t *) xb, (dist_t *) (xb + lb), scratch_counts, nullptr, nullptr);
	float kms = 0.f;
	if (rc == HNSW_GPU_OK && hnsw_gpu_ctx_search_ms(ctx, 0, &kms) == HNSW_GPU_OK) g_cnt.kernel_ns += (uint64_t) (kms * 1e6f);   // (waits for it)
	// every peer's answer (also after a failure: a connection must not be left with a response in flight)
	for (size_t k = 0; k < sf.fds.size(); k++)
	{
		if (sf.fds[k] < 0) continue;
		hgs_hdr resp;
		int got = -1;
		if (hgs::recv_exact(sf.fds[k], &resp, sizeof(resp), &got, 60000) != 0 || resp.magic != HGS_MAGIC || resp.op != HGS_OP_SHARD_SEARCH || resp.len != 0)
		{
			if (got >= 0) close(got);
			close(sf.fds[k]); sf.fds[k] = -1; sf.attached[k] = false;
			logf("sharded batch: shard peer %s did not answer", g_opt.shard_peers[k].c_str());
			if (rc == HNSW_GPU_OK) rc = HGS_ERR_IO;
			continue;
		}
		if (got >= 0) close(got);
		if (resp.status != HGS_OK && rc == HNSW_GPU_OK)
		{
			logf("sharded batch: shard peer %s answered %d", g_opt.shard_peers[k].c_str(), (int) resp.status);
			rc = resp.status;
		}
	}
	if (rc != HNSW_GPU_OK) return rc;
	// the exchange is done: every list is in this device's memory.  Merge into the pinned outputs and wait for it.
	rc = hnsw_gpu_merge_topk_strided_dev(g_opt.device, (const label_t *) xb, nq * ef, (const dist_t *) (xb + lb), nq * ef, lists, nq, ef, L, D, C, nullptr);
	if (rc == HNSW_GPU_OK) rc = hnsw_gpu_device_wait(g_opt.device, nullptr);
	return rc;
}

void run_batch(int d, std::vector<SReq> &batch, Pinned &pin)
{
	Entry *e = batch[0].e.get();
	const size_t nq = batch.size(), ef = batch[0].h.aux, dim = e->meta.dim;
	const size_t qb = (nq * dim * 4 + 255) & ~(size_t) 255, lb = (nq * ef * 8 + 255) & ~(size_t) 255,
				 db = (nq * ef * 4 + 255) & ~(size_t) 255, cb = (nq * 4 + 255) & ~(size_t) 255;
	const bool sharded = !g_opt.shard_peers.empty();
	int rc = HNSW_GPU_OK;
	if (!pin.reserve(qb + lb + db + cb + (sharded ? cb : 0))) rc = HNSW_GPU_ERR_NOMEM;
	char *base = (char *) pin.p;
	float *Q = (float *) base;
	label_t *L = (label_t *) (base + qb);
	dist_t *D = (dist_t *) (base + qb + lb);
	uint32_t *C = (uint32_t *) (base + qb + lb + db);
	const uint64_t t0 = now_ns();
	if (rc == HNSW_GPU_OK)
	{
		for (size_t i = 0; i < nq; i++) memcpy(Q + i * dim, batch[i].q.data(), dim * 4);
		e->begin_read();
		if (sharded)                                             // this server is the front of a row-sharded index: every shard, then the merge
			rc = run_batch_sharded(d, e, nq, ef, Q, L, D, C, (uint32_t *) (base + qb + lb + db + cb));
		else
		{
			hnsw_gpu_ctx *ctx = e->context(d);
			if (!ctx) rc = HNSW_GPU_ERR_HIP;
			else rc = hnsw_gpu_search_batch_ctx_host(ctx, Q, nq, ef, L, D, C);
			float kms = 0.f;
			if (rc == HNSW_GPU_OK && hnsw_gpu_ctx_search_ms(ctx, 0, &kms) == HNSW_GPU_OK) g_cnt.kernel_ns += (uint64_t) (kms * 1e6f);
		}
		e->end_read();
		if (rc != HNSW_GPU_OK) logf("search batch of %zu (ef %zu) failed: %s", nq, ef, hnsw_gpu_last_error());
	}
	e->last_used.store(now_ns());
	g_cnt.batch_ns += now_ns() - t0;
	g_cnt.batches++;
	g_cnt.searches += nq;
	uint64_t mb = g_cnt.max_batch.load();
	while (nq > mb && !g_cnt.max_batch.compare_exchange_weak(mb, nq)) {}
	if (rc != HNSW_GPU_OK) g_cnt.search_errors += nq;
	for (size_t i = 0; i < nq; i++)
	{
		SReq &r = batch[i];
		if (rc != HNSW_GPU_OK) { r.c->respond(r.h, rc); continue; }
		const size_t cnt = C[i];
		if (cnt > ef)                                          // never a result count (an interrupted launch marks unanswered queries 0xFFFFFFFF)
		{
			g_cnt.search_errors++;
			r.c->respond(r.h, HNSW_GPU_ERR_INTERNAL);
			continue;
		}
		r.c->respond(r.h, HGS_OK, cnt, 0, L + i * ef, cnt * 8, r.h.a0 ? D + i * ef : nullptr, cnt * 4, e->gen.load());
	}
}

// Take every queued SEARCH for the (mirror, ef) of the oldest one, up to --max-batch.  g_q_mu held.
void take_batch(std::vector<SReq> &batch)
{
	Entry *e = g_q.front().e.get();
	const uint32_t ef = g_q.front().h.aux;
	for (auto it = g_q.begin(); it != g_q.end() && batch.size() < g_opt.max_batch;)
	{
		if (it->e.get() == e && it->h.aux == ef) { batch.push_back(std::move(*it)); it = g_q.erase(it); }
		else ++it;
	}
}

void answer_leftovers()
{
	std::lock_guard<std::mutex> lk(g_q_mu);
	for (SReq &r : g_q) r.c->respond(r.h, HGS_ERR_SHUTDOWN);
	g_q.clear();
}

// --lanes 0: one blocking launch at a time per dispatcher; every answer of a batch leaves when the
// slowest query of the batch is done.
void dispatcher_blocking(int d)
{
	Pinned pin;
	std::vector<SReq> batch;
	while (true)
	{
		batch.clear();
		{
			std::unique_lock<std::mutex> lk(g_q_mu);
			g_q_cv.wait(lk, [] { return g_stop.load() || !g_q.empty(); });
			if (g_stop.load()) break;
			if (g_opt.linger_us > 0 && g_q.size() < g_opt.min_batch)
				HGS_TIMED_WAIT(g_q_cv, lk, std::chrono::microseconds(g_opt.linger_us),
								[] { return g_stop.load() || g_q.size() >= g_opt.min_batch; });
			if (g_q.empty()) continue;
			take_batch(batch);
		}
		run_batch(d, batch, pin);
	}
	answer_leftovers();
}

// Streamed completion (default).  A lane = one launch in flight: its queries, result arrays and
// completion flags live in pinned host memory that the kernel reads and writes directly
// (hnsw_gpu_search_batch_ctx_flags), so nothing is copied and nothing is waited for: the dispatcher
// polls the flags and answers each backend the moment ITS walk has ended — not when the slowest of
// the batch has — while new requests go out on the next free lane.
struct Lane
{
	Pinned pin;
	std::vector<SReq> batch;
	std::vector<uint32_t> pending;           // positions in `batch` not answered yet
	EntryP e;
	hnsw_gpu_ctx *ctx = nullptr;
	size_t ef = 0;
	label_t *L = nullptr; dist_t *D = nullptr; uint32_t *C = nullptr;
	volatile uint32_t *F = nullptr;
	uint64_t t0 = 0, t_check = 0;
	bool active = false;
};

const uint64_t LANE_TIMEOUT_NS = 60ull * 1000000000ull;

bool lane_launch(int ctx_slot, Lane &ln)
{
	Entry *e = ln.batch[0].e.get();
	if (!e->try_read())                      // a writer is waiting or working: put the requests back, in order
	{
		std::lock_guard<std::mutex> lk(g_q_mu);
		for (size_t i = ln.batch.size(); i-- > 0;) g_q.push_front(std::move(ln.batch[i]));
		ln.batch.clear();
		return false;
	}
	const size_t nq = ln.batch.size(), ef = ln.batch[0].h.aux, dim = e->meta.dim;
	const size_t qb = (nq * dim * 4 + 255) & ~(size_t) 255, lb = (nq * ef * 8 + 255) & ~(size_t) 255,
				 db = (nq * ef * 4 + 255) & ~(size_t) 255, cb = (nq * 4 + 255) & ~(size_t) 255, fb = nq * 4;
	int rc = ln.pin.reserve(qb + lb + db + cb + fb) ? HNSW_GPU_OK : HNSW_GPU_ERR_NOMEM;
	if (rc == HNSW_GPU_OK)
	{
		char *base = (char *) ln.pin.p;
		float *Q = (float *) base;
		ln.L = (label_t *) (base + qb);
		ln.D = (dist_t *) (base + qb + lb);
		ln.C = (uint32_t *) (base + qb + lb + db);
		ln.F = (volatile uint32_t *) (base + qb + lb + db + cb);
		for (size_t i = 0; i < nq; i++) memcpy(Q + i * dim, ln.batch[i].q.data(), dim * 4);
		memset((void *) ln.F, 0, fb);
		ln.ctx = e->context(ctx_slot);
		if (ln.ctx)
		{
			unsigned walkers = g_opt.walkers < 0 ? 0u : (unsigned) g_opt.walkers;
			if (g_opt.walkers < 0)
			{
				const long w = g_walks_in_flight.load(std::memory_order_relaxed) + (long) nq;
				walkers = (unsigned) std::min<long>(8, std::max<long>(1, (w + g_device_blocks - 1) / g_device_blocks));
			}
			(void) hnsw_gpu_ctx_set_walkers(ln.ctx, walkers);
		}
		rc = ln.ctx ? hnsw_gpu_search_batch_ctx_flags(ln.ctx, Q, nq, ef, ln.L, ln.D, ln.C, nullptr, (uint32_t *) ln.F)
					: HNSW_GPU_ERR_HIP;
	}
	if (rc != HNSW_GPU_OK)
	{
		logf("search launch of %zu (ef %zu) failed: %s", nq, ef, hnsw_gpu_last_error());
		e->end_read();
		g_cnt.search_errors += nq;
		g_cnt.searches += nq;
		g_cnt.batches++;
		for (SReq &r : ln.batch) r.c->respond(r.h, rc);
		ln.batch.clear();
		return false;
	}
	ln.e = ln.batch[0].e;
	ln.ef = ef;
	ln.pending.resize(nq);
	for (size_t i = 0; i < nq; i++) ln.pending[i] = (uint32_t) i;
	ln.t0 = ln.t_check = now_ns();
	ln.active = true;
	{
		uint64_t waited = 0;
		for (const SReq &r : ln.batch) waited += ln.t0 - std::min(ln.t0, r.t_in);
		g_cnt.queue_ns += waited;
	}
	g_walks_in_flight.fetch_add((long) nq, std::memory_order_relaxed);
	g_cnt.batches++;
	g_cnt.searches += nq;
	uint64_t mb = g_cnt.max_batch.load();
	while (nq > mb && !g_cnt.max_batch.compare_exchange_weak(mb, nq)) {}
	return true;
}

// Answer what has completed; retire the lane when everything has.  True if anything happened.
bool lane_poll(Lane &ln)
{
	bool progress = false;
	const uint64_t gen = ln.e->gen.load();
	for (size_t k = 0; k < ln.pending.size();)
	{
		const uint32_t i = ln.pending[k];
		if (__atomic_load_n((const uint32_t *) &ln.F[i], __ATOMIC_ACQUIRE))
		{
			SReq &r = ln.batch[i];
			const size_t cnt = ln.C[i] <= ln.ef ? ln.C[i] : 0;
			const uint64_t t_seen = now_ns();
			r.c->respond(r.h, HGS_OK, cnt, 0, ln.L + (size_t) i * ln.ef, cnt * 8, r.h.a0 ? ln.D + (size_t) i * ln.ef : nullptr, cnt * 4,
						 gen);
			g_cnt.walk_ns += t_seen - ln.t0;
			g_cnt.answer_ns += now_ns() - t_seen;
			ln.pending[k] = ln.pending.back();
			ln.pending.pop_back();
			g_walks_in_flight.fetch_sub(1, std::memory_order_relaxed);
			progress = true;
		}
		else k++;
	}
	// Has the launch left the device?  Asked when every flag is in, and every 2 ms as the failure detector.
	int idle = 0;
	const uint64_t now = now_ns();
	if (ln.pending.empty() || now - ln.t_check > 2000000ull)
	{
		idle = hnsw_gpu_ctx_idle(ln.ctx);
		ln.t_check = now;
	}
	bool fail = idle < 0 || (idle == 1 && !ln.pending.empty() && [&] {
		// the launch has left the device: every flag is visible by now — look once more before giving up
		for (uint32_t i : ln.pending) if (!__atomic_load_n((const uint32_t *) &ln.F[i], __ATOMIC_ACQUIRE)) return true;
		return false;
	}());
	if (!fail && !ln.pending.empty() && now_ns() - ln.t0 > LANE_TIMEOUT_NS) fail = true;
	if (fail)
	{
		logf("search launch lost %zu of %zu queries: %s", ln.pending.size(), ln.batch.size(), hnsw_gpu_last_error());
		g_cnt.search_errors += ln.pending.size();
		for (uint32_t i : ln.pending) ln.batch[i].c->respond(ln.batch[i].h, HNSW_GPU_ERR_INTERNAL);
		g_walks_in_flight.fetch_sub((long) ln.pending.size(), std::memory_order_relaxed);
		ln.pending.clear();
		progress = true;
	}
	if (ln.pending.empty() && (idle == 1 || fail))
	{
		float kms = 0.f;
		if (!fail && hnsw_gpu_ctx_search_ms(ln.ctx, 0, &kms) == HNSW_GPU_OK) g_cnt.kernel_ns += (uint64_t) (kms * 1e6f);
		g_cnt.batch_ns += now_ns() - ln.t0;
		ln.e->last_used.store(now_ns());
		ln.e->end_read();
		ln.e.reset();
		ln.batch.clear();
		ln.active = false;
		progress = true;
	}
	return progress;
}

void dispatcher_lanes(int d)
{
	std::vector<Lane> lanes((size_t) g_opt.lanes);
	unsigned idle_spins = 0;
	while (!g_stop.load())
	{
		bool progress = false;
		size_t active = 0;
		for (Lane &ln : lanes) active += ln.active ? 1 : 0;
		for (size_t li = 0; li < lanes.size(); li++)
		{
			Lane &ln = lanes[li];
			if (ln.active) continue;
			{
				std::unique_lock<std::mutex> lk(g_q_mu);
				if (active == 0)           // nothing in flight here: sleep until there is work
				{
					g_q_cv.wait(lk, [] { return g_stop.load() || !g_q.empty(); });
					if (g_stop.load()) break;
					if (g_opt.linger_us > 0 && g_q.size() < g_opt.min_batch)
						HGS_TIMED_WAIT(g_q_cv, lk, std::chrono::microseconds(g_opt.linger_us),
										[] { return g_stop.load() || g_q.size() >= g_opt.min_batch; });
				}
				if (g_q.empty()) break;
				take_batch(ln.batch);
			}
			if (lane_launch(d * g_opt.lanes + (int) li, ln)) { active++; progress = true; }
			else if (active == 0) std::this_thread::sleep_for(std::chrono::microseconds(50));   // writer at work
			break;                         // poll before filling another lane: requests gather meanwhile
		}
		for (Lane &ln : lanes)
			if (ln.active && lane_poll(ln)) progress = true;
		if (progress) idle_spins = 0;
		else if (++idle_spins > 64) { std::this_thread::yield(); }
		else __builtin_ia32_pause();
	}
	// stopping: let what is in flight finish (bounded), then refuse the rest
	const uint64_t t_end = now_ns() + 5ull * 1000000000ull;
	for (Lane &ln : lanes)
		while (ln.active && now_ns() < t_end)
			if (!lane_poll(ln)) std::this_thread::yield();
	for (Lane &ln : lanes)
		if (ln.active)
		{
			for (uint32_t i : ln.pending) ln.batch[i].c->respond(ln.batch[i].h, HGS_ERR_SHUTDOWN);
			g_walks_in_flight.fetch_sub((long) ln.pending.size(), std::memory_order_relaxed);
			ln.e->end_read();
			ln.active = false;
		}
	answer_leftovers();
}

void dispatcher_main(int d)
{
	pthread_setname_np(pthread_self(), "hgs-dispatch");
	if (g_opt.lanes > 0) dispatcher_lanes(d);
	else dispatcher_blocking(d);
}


// ----------------------------------------------------------------------------- stream mode (--stream 1)
// One resident search launch per (mirror, ef) at a time, fed through a ring in pinned host memory (include/hnsw_gpu.h, "Streams"):
// a SEARCH is written into the next ring slot by the READER thread that parsed it and published with one store; the kernel's
// walking waves pick queries up as they become free, the answer threads poll the ring's completion flags and answer each backend
// the moment its walk has ended.  No batch is formed, nothing waits for a launch to end: the 0.4-0.5 ms a query spent waiting for a
// lane at 1 024 backends (profiles/r4f_server_walkers_and_breakdown.txt) is gone.  The session gives way — stops accepting, lets its
// walks finish, closes — when a writer wants the mirror, when searches for another (mirror, ef) are waiting, when nothing has been
// outstanding for a while (a resident launch holds the whole device), when the load calls for another team geometry, at shutdown.
// Who is inside Session::submit right now: one mark per producer thread (a cache line each: nobody shares one), set to the session on the
// way in and cleared on the way out.  close_session stores accepting = false and then waits until no mark names the session — both sides
// sequentially consistent, so a producer either sees the session closing and leaves, or the closer sees its mark and waits: nobody touches
// the ring (or calls hnsw_gpu_stream_publish) after hnsw_gpu_stream_close has freed it.  Two shared counters would do the same at the
// price of two more contended read-modify-writes per search.
struct alignas(64) InSubmit { std::atomic<const void *> s{nullptr}; };
constexpr int MAX_PRODUCERS = 256;
InSubmit g_in_submit[MAX_PRODUCERS];
std::atomic<int> g_producers{0};
InSubmit &my_submit_mark()
{
	static thread_local InSubmit *mine = nullptr;
	if (!mine) mine = &g_in_submit[std::min(g_producers.fetch_add(1), MAX_PRODUCERS - 1)];   // (readers + the manager: far fewer than 256)
	return *mine;
}

struct Session
{
	EntryP e;
	size_t ef = 0, dim = 0;
	hnsw_gpu_ctx *ctx = nullptr;
	hnsw_gpu_stream *st = nullptr;
	float *Q = nullptr; label_t *L = nullptr; dist_t *D = nullptr; uint32_t *C = nullptr; volatile uint32_t *F = nullptr;
	uint32_t ring = 0;
	// producers (the reader threads) share no lock: a ticket is claimed with one atomic add, its slot is written by its owner alone,
	// and the published count is advanced over the tickets that are ready IN ORDER by whoever gets there (a mutex around "ticket +
	// 3 KB copy + publish" capped the whole server at one critical section per 2.3 us = 0.44 M q/s, profiles/r4i_*)
	std::atomic<uint32_t> claim{0}, pub{0};
	std::unique_ptr<std::atomic<uint32_t>[]> ready;   // per slot: ticket + 1 once its query is written
	unsigned walkers = 0;
	long capacity = 0;                      // walks the launch runs at once
	std::vector<SReq> req;                  // per slot
	std::vector<uint64_t> t_pub;            // per slot: when it was published
	std::unique_ptr<std::atomic<uint8_t>[]> busy;   // per slot: 1 = published, not answered yet
	std::atomic<long> outstanding{0};
	std::atomic<uint32_t> bailed{0};        // tickets a producer claimed and then gave up (submit, closing session): never ready, so `pub` stops in front of them
	std::atomic<bool> accepting{false};
	std::atomic<uint64_t> t_active{0};      // last publish or answer
	std::atomic<uint64_t> t_crowded{0}, t_roomy{0};   // since when the launch has been too small / far too large for the load (0 = it is not)

	// reader threads.  false = not taken (the caller queues the request; the manager sees to it)
	bool submit(SReq &r)
	{
		if (!accepting.load(std::memory_order_acquire)) return false;
		struct Mark
		{
			InSubmit &m;
			Mark(InSubmit &mm, const void *who) : m(mm) { m.s.store(who); }                       // (sequentially consistent)
			~Mark() { m.s.store(nullptr, std::memory_order_release); }
		} inside(my_submit_mark(), this);
		if (!accepting.load()) return false;                                                      // closing: the closer may not have seen my mark
		// (room: a backend has one search outstanding, the ring has several slots per backend; a full ring means stragglers a whole
		// ring old — the request waits in the queue instead)
		if (outstanding.load(std::memory_order_relaxed) >= (long) ring - 64) return false;
		outstanding.fetch_add(1, std::memory_order_acq_rel);
		const uint32_t t = claim.fetch_add(1, std::memory_order_acq_rel);
		const uint32_t slot = t & (ring - 1);
		for (unsigned spin = 0; busy[slot].load(std::memory_order_acquire); spin++)     // its occupant of a ring ago is being answered right now
		{
			// ... or is a walk that hangs: then the session is closed over it (stream_manager_main), and this request queues instead of
			// spinning inside a ring that is going away.  The ticket stays unpublished; the closer does not wait for it for ever.
			if (spin > 1000 && !accepting.load(std::memory_order_acquire))
			{
				bailed.fetch_add(1);                                               // (the closer stops waiting for `pub` to reach `claim`: close_session)
				outstanding.fetch_sub(1, std::memory_order_acq_rel);
				return false;
			}
			if (spin > 1000) std::this_thread::yield(); else __builtin_ia32_pause();
		}
		memcpy(Q + (size_t) slot * dim, r.q.data(), dim * 4);
		F[slot] = 0;
		const uint64_t now = now_ns();
		g_cnt.queue_ns += now - std::min(now, r.t_in);
		t_pub[slot] = now;
		req[slot] = std::move(r);
		busy[slot].store(1, std::memory_order_release);
		// SEQUENTIALLY CONSISTENT, and so are the looks in advance(): "store my ticket's ready word, then look at my neighbour's" on two
		// producers at once is the store-buffer pattern — with release / acquire BOTH may miss the other's store, both stop, and the later
		// ticket stays unpublished until another producer comes by.  In a busy session that is a blip; at a session's close nobody comes
		// by any more and the query was answered with an error after the close's patience ran out ("stream closed with 1 queries
		// unanswered": twice at 2 048 backends on the device, once on the CPU tier's double — profiles/r4aj_mailboxes.txt blamed the
		// capped host for it, wrongly).  The closing manager also advances by itself (close_session).
		ready[slot].store(t + 1);
		advance();
		t_active.store(now, std::memory_order_relaxed);
		return true;
	}
	// advance the published count over every ticket that is ready, in order; whoever is behind an unready ticket leaves the rest to
	// that ticket's owner (it runs this after its own store)
	void advance()
	{
		uint32_t p = pub.load();
		bool moved = false;
		while (ready[p & (ring - 1)].load() == p + 1)
			if (pub.compare_exchange_weak(p, p + 1)) { p = p + 1; moved = true; }
		if (moved) (void) hnsw_gpu_stream_publish(st, p);                  // (the library keeps the maximum: two advancers may arrive out of order)
	}
	// has the ticket that sits in `slot` been published (handed to the launch)?  (its ready word holds ticket + 1)
	bool published(uint32_t slot) const { return (int32_t) (pub.load() - (ready[slot].load() - 1u)) > 0; }
	// walks the launch still owes an answer: busy slots whose ticket was published
	long walks_owed() const
	{
		long n = 0;
		for (uint32_t slot = 0; slot < ring; slot++)
			if (busy[slot].load(std::memory_order_acquire) && published(slot)) n++;
		return n;
	}
};
using SessionP = std::shared_ptr<Session>;
std::mutex g_sess_mu;
SessionP g_sess;                            // the open session, if any (g_sess_mu)
std::atomic<uint64_t> g_sess_gen{0};        // bumped at every change of g_sess: the hot paths keep a thread-local copy and look at this word only

constexpr int MAX_ANSWER_THREADS = 64;      // (main() refuses --stream 1 with more dispatchers)
std::atomic<uint64_t> g_answer_gen[MAX_ANSWER_THREADS];     // per answer thread: the session generation it has refreshed to
std::atomic<bool> g_answer_exited[MAX_ANSWER_THREADS];      // per answer thread: it has returned (it holds nothing any more)
std::atomic<bool> g_sessions_over{false};   // shutdown: the manager has closed the last session — only then do the answer threads leave

void set_session(const SessionP &ss)
{
	std::lock_guard<std::mutex> lk(g_sess_mu);
	g_sess = ss;
	g_sess_gen.fetch_add(1, std::memory_order_release);
}

SessionP current_session()
{
	static thread_local SessionP mine;
	static thread_local uint64_t seen = ~0ull;
	const uint64_t gen = g_sess_gen.load(std::memory_order_acquire);
	if (gen != seen)
	{
		std::lock_guard<std::mutex> lk(g_sess_mu);
		mine = g_sess;
		seen = g_sess_gen.load(std::memory_order_relaxed);
	}
	return mine;
}

// answer thread k of n: its stripe of the ring
void stream_answer_main(int k, int n)
{
	pthread_setname_np(pthread_self(), "hgs-answer");
	unsigned idle = 0;
	// (until the manager has closed the last session, not just until g_stop: the walks that are outstanding at shutdown are answered,
	// not left to the close's patience)
	while (!g_sessions_over.load(std::memory_order_acquire))
	{
		SessionP ss = current_session();
		g_answer_gen[k].store(g_sess_gen.load(std::memory_order_acquire), std::memory_order_release);   // "I hold nothing older than this"
		if (!ss) { std::this_thread::sleep_for(std::chrono::microseconds(50)); continue; }
		bool progress = false;
		const uint64_t gen = ss->e->gen.load();
		for (uint32_t slot = (uint32_t) k; slot < ss->ring; slot += (uint32_t) n)
		{
			if (!ss->busy[slot].load(std::memory_order_acquire)) continue;
			if (!__atomic_load_n((const uint32_t *) &ss->F[slot], __ATOMIC_ACQUIRE)) continue;
			const uint64_t t_seen = now_ns();
			SReq r = std::move(ss->req[slot]);
			const size_t ef = ss->ef;
			const uint32_t c = ss->C[slot];
			const size_t cnt = c <= ef ? c : 0;
			g_cnt.searches++;                                   // (before the answer leaves: a backend that asks for STATS next finds itself counted)
			if (c > ef) { g_cnt.search_errors++; r.c->respond(r.h, HNSW_GPU_ERR_INTERNAL); }
			else r.c->respond(r.h, HGS_OK, cnt, 0, ss->L + (size_t) slot * ef, cnt * 8, r.h.a0 ? ss->D + (size_t) slot * ef : nullptr, cnt * 4, gen);
			g_cnt.walk_ns += t_seen - ss->t_pub[slot];
			g_cnt.answer_ns += now_ns() - t_seen;
			ss->busy[slot].store(0, std::memory_order_release);
			ss->outstanding.fetch_sub(1, std::memory_order_relaxed);
			ss->t_active.store(t_seen, std::memory_order_relaxed);
			progress = true;
		}
		if (progress) idle = 0;
		else if (++idle > 256) std::this_thread::yield();
		else __builtin_ia32_pause();
	}
	g_answer_exited[k].store(true);
}

// Close the session: no new queries, let the walks in flight finish (bounded), stop the launch, release the mirror.
void close_session(SessionP &ss, const char *why, int unanswered_rc = HNSW_GPU_ERR_INTERNAL)
{
	ss->accepting.store(false);             // (sequentially consistent, as the producers' marks: Session::submit)
	const uint64_t t0 = now_ns();
	bool somebody_inside = false;           // a thread of ours may still touch the ring: it is then given up, not freed
	for (int i = 0, n = std::min(g_producers.load(), MAX_PRODUCERS); i < n; i++)        // producers inside submit finish their slot and leave
	{
		while (g_in_submit[i].s.load() == (const void *) ss.get() && now_ns() - t0 < 1000000000ull)
			std::this_thread::yield();
		if (g_in_submit[i].s.load() == (const void *) ss.get()) somebody_inside = true;
	}
	// producers that were past the check finish their slot; then every claimed ticket is published and, walked, answered
	// (2 s of patience: on a host whose CPU time is capped the whole process may be frozen for tens of milliseconds at a time)
	// A producer that gave its ticket up (Session::submit, bail-out) leaves a hole `pub` cannot cross: from then on the drain is over when
	// every PUBLISHED walk is answered — what sits behind the hole was never handed to the launch and goes back to the queue below
	// (such a close used to wait its whole 2 s and then answer those requests with an error).
	while (now_ns() - t0 < 2000000000ull)
	{
		ss->advance();                      // (no producer will come by any more: a ready ticket that two of them left to each other is published here)
		if (ss->bailed.load() == 0 ? (ss->outstanding.load() <= 0 && ss->pub.load() == ss->claim.load()) : ss->walks_owed() == 0) break;
		std::this_thread::sleep_for(std::chrono::microseconds(20));
	}
	const uint64_t t_drained = now_ns();
	if (t_drained - t0 > 200000000ull)      // (a close is a millisecond or two: say so when it is not, with what it was waiting for)
		logf("slow close (%s): %.0f ms until the ring had drained — outstanding %ld, claimed %u, published %u", why, (t_drained - t0) / 1e6,
			 ss->outstanding.load(), ss->claim.load(), ss->pub.load());
	set_session(nullptr);                   // the answer threads and readers let go of it at their next look ...
	// ... (their thread-local copies: a reader that is idle keeps one until its next request, so the count cannot be waited on; what
	// matters is that nobody is INSIDE the ring: producers have left (their marks, above) and none enters any more, and the answer
	// threads have refreshed once every one of them has passed the generation check)
	{
		const uint64_t t1 = now_ns(), gen = g_sess_gen.load();
		for (int k = 0; k < g_opt.dispatchers && k < MAX_ANSWER_THREADS; k++)
		{
			while (g_answer_gen[k].load(std::memory_order_acquire) < gen && !g_answer_exited[k].load() && now_ns() - t1 < 1000000000ull)
				std::this_thread::sleep_for(std::chrono::microseconds(20));
			if (g_answer_gen[k].load(std::memory_order_acquire) < gen && !g_answer_exited[k].load()) somebody_inside = true;
		}
	}
	const uint64_t t_let_go = now_ns();
	if (somebody_inside) logf("closing the stream (%s): a thread did not leave the ring within its second: the ring is given up, not freed", why);
	const int rc = somebody_inside ? hnsw_gpu_stream_abandon(ss->st) : hnsw_gpu_stream_close(ss->st);
	if (rc != HNSW_GPU_OK) logf("closing the stream: %s", hnsw_gpu_last_error());
	if (now_ns() - t_drained > 200000000ull)
		logf("slow close (%s): %.0f ms for the answer threads to let go, %.0f ms for the launch to end", why, (t_let_go - t_drained) / 1e6,
			 (now_ns() - t_let_go) / 1e6);
	long lost = 0, requeued = 0;
	// requests that were written into the ring but never published (behind a bailed ticket): nobody has walked them and nobody will —
	// back to the head of the queue, oldest ticket first, unless the server is going down
	if (unanswered_rc != HGS_ERR_SHUTDOWN)
	{
		std::vector<std::pair<uint32_t, uint32_t>> back;     // (ticket, slot)
		for (uint32_t slot = 0; slot < ss->ring; slot++)
			if (ss->busy[slot].load() && !ss->published(slot) && !ss->F[slot]) back.emplace_back(ss->ready[slot].load() - 1u, slot);
		std::sort(back.begin(), back.end(), [&](const std::pair<uint32_t, uint32_t> &a, const std::pair<uint32_t, uint32_t> &b) { return (int32_t) (a.first - b.first) > 0; });
		if (!back.empty())
		{
			std::lock_guard<std::mutex> lk(g_q_mu);
			for (auto &tb : back)                           // (newest first, each to the front: the oldest ends up first)
			{
				g_q.push_front(std::move(ss->req[tb.second]));
				ss->busy[tb.second].store(0, std::memory_order_release);
				ss->outstanding.fetch_sub(1, std::memory_order_relaxed);
				requeued++;
			}
			g_q_waiting.store((long) g_q.size(), std::memory_order_release);
		}
		if (requeued) { logf("stream closed (%s): %ld requests behind a ticket its producer gave up go back to the queue", why, requeued); g_q_cv.notify_all(); }
	}
	for (uint32_t slot = 0; slot < ss->ring; slot++)
		if (ss->busy[slot].load())
		{
			// (what such a slot looked like, for whoever has to find out why: a query that was never published shows ready != its ticket
			// or published <= its ticket; one that was published and not walked shows flag 0; one that was walked and not answered flag 1)
			logf("unanswered slot %u: flag %u, ready word %u, published %u, claimed %u, outstanding %ld, in the ring for %.1f ms, the drain took %.0f ms",
				 slot, (unsigned) ss->F[slot], ss->ready[slot].load(), ss->pub.load(), ss->claim.load(), ss->outstanding.load(),
				 (now_ns() - ss->t_pub[slot]) / 1e6, (t_drained - t0) / 1e6);
			ss->req[slot].c->respond(ss->req[slot].h, unanswered_rc);
			ss->busy[slot].store(0, std::memory_order_release);
			lost++;
		}
	if (lost) { g_cnt.search_errors += (uint64_t) lost; logf("stream closed (%s) with %ld queries unanswered", why, lost); }
	VLOG("stream on %llu closed: %s (%u queries)", (unsigned long long) ss->e->key, why, ss->pub.load());
	ss->e->last_used.store(now_ns());
	ss->e->end_read();
	g_cnt.batches++;
}

SessionP open_session(const EntryP &e, size_t ef, size_t backlog)
{
	SessionP ss = std::make_shared<Session>();
	ss->e = e; ss->ef = ef; ss->dim = e->meta.dim;
	ss->ring = (uint32_t) g_opt.ring;
	ss->ctx = e->context(0);
	if (!ss->ctx) return nullptr;
	// team geometry by the load that is waiting (g_device_blocks - 1 walking blocks: block 0 is the doorbell)
	const long blocks = std::max(1, g_device_blocks - 1);
	unsigned walkers = g_opt.walkers > 0 ? (unsigned) g_opt.walkers
										  : (unsigned) std::min<long>(8, std::max<long>(1, ((long) backlog + blocks - 1) / blocks));
	if (hnsw_gpu_stream_open(ss->ctx, ef, ss->ring, walkers, &ss->st) != HNSW_GPU_OK) return nullptr;
	uint32_t *f = nullptr;
	(void) hnsw_gpu_stream_buffers(ss->st, &ss->Q, &ss->L, &ss->D, &ss->C, &f);
	ss->F = f;
	ss->walkers = walkers;
	ss->capacity = blocks * (long) walkers;
	ss->req.resize(ss->ring);
	ss->t_pub.assign(ss->ring, 0);
	ss->busy.reset(new std::atomic<uint8_t>[ss->ring]);
	ss->ready.reset(new std::atomic<uint32_t>[ss->ring]);
	for (uint32_t i = 0; i < ss->ring; i++) { ss->busy[i].store(0); ss->ready[i].store(0); }
	ss->t_active.store(now_ns());
	ss->accepting.store(true);
	VLOG("stream on %llu opened: ef %zu, %u walking waves per block (backlog %zu)", (unsigned long long) e->key, ef, walkers, backlog);
	return ss;
}

const uint64_t STREAM_WALK_TIMEOUT_NS = 60ull * 1000000000ull;   // (as LANE_TIMEOUT_NS of the lanes: a walk is a millisecond)

void stream_manager_main()
{
	pthread_setname_np(pthread_self(), "hgs-manager");
	Pinned pin;
	std::vector<SReq> batch;
	const uint64_t IDLE_NS = 2000000ull, CROWDED_NS = 2000000ull, ROOMY_NS = 100000000ull;
	while (!g_stop.load())
	{
		SessionP ss = current_session();
		if (ss)
		{
			const uint64_t now = now_ns();
			const long out = ss->outstanding.load();
			// too small / far too large a launch for the load?  (hysteresis: a change costs one drain)
			if (g_opt.walkers <= 0)
			{
				// (walking waves per block = just enough for the walks in flight: every further one is a helper less for somebody)
				if (out > ss->capacity - ss->capacity / 16 && ss->walkers < 8) { uint64_t z = 0; ss->t_crowded.compare_exchange_strong(z, now); }
				else ss->t_crowded.store(0);
				const long blocks = std::max(1, g_device_blocks - 1);
				if (ss->walkers > 1 && out * 5 / 4 <= blocks * (long) (ss->walkers - 1)) { uint64_t z = 0; ss->t_roomy.compare_exchange_strong(z, now); }
				else ss->t_roomy.store(0);
			}
			const char *why = nullptr;
			size_t hint = 0;
			// liveness of the resident launch (the library's watchdog does not cover it): a launch that has left the device (abort, fault)
			// or a walk that has been in the ring longer than any walk takes — checked every 100 ms — ends the session; close_session
			// answers what is unanswered with an error and the next request opens a new one
			static thread_local uint64_t t_scan = 0;
			bool dead = false, overdue = false;
			if (now - t_scan > 100000000ull)
			{
				t_scan = now;
				dead = hnsw_gpu_stream_alive(ss->st) != 1;
				if (!dead && out > 0)
					for (uint32_t slot = 0; slot < ss->ring && !overdue; slot++)
						if (ss->busy[slot].load(std::memory_order_acquire) && now > ss->t_pub[slot] && now - ss->t_pub[slot] > STREAM_WALK_TIMEOUT_NS)
							overdue = true;
			}
			if (dead) why = "the resident launch has left the device";
			else if (overdue) why = "a walk has been in the ring for longer than the time-out";
			else if (ss->e->writer_wants()) why = "a writer wants the mirror";
			else if (g_device_wanted.load() > 0) why = "control work needs the device";
			else if (out == 0 && now - ss->t_active.load() > IDLE_NS) why = "idle";
			else if (ss->t_crowded.load() && now - ss->t_crowded.load() > CROWDED_NS) { why = "more walking waves per block needed"; hint = (size_t) (out + out / 8 + 1); }
			else if (ss->t_roomy.load() && now - ss->t_roomy.load() > ROOMY_NS) { why = "fewer walking waves per block suffice"; hint = (size_t) std::max<long>(out + out / 4, 1); }
			else
			{
				// requests that did not get in by themselves (arrived between sessions, ring full): in order; another (mirror, ef) ends the session
				std::unique_lock<std::mutex> lk(g_q_mu);
				while (!g_q.empty())
				{
					SReq &r = g_q.front();
					if (r.e.get() != ss->e.get() || r.h.aux != ss->ef) { why = "searches for another mirror or beam are waiting"; break; }
					if (!ss->submit(r)) break;
					g_q.pop_front();
				}
				g_q_waiting.store((long) g_q.size(), std::memory_order_release);
			}
			if (why)
			{
				close_session(ss, why);
				if (hint && !g_stop.load() && !ss->e->writer_wants() && ss->e->try_read())
				{
					SessionP nn = open_session(ss->e, ss->ef, hint);
					if (nn) set_session(nn);
					else ss->e->end_read();
				}
				continue;
			}
			ss->advance();                  // (bounds the blip of a ready ticket that two producers left to each other: Session::submit)
			if (g_q_waiting.load(std::memory_order_acquire) == 0) std::this_thread::sleep_for(std::chrono::microseconds(50));
			continue;
		}
		// no session: wait for work, open one for the oldest request's (mirror, ef)
		if (g_device_wanted.load() > 0)
		{
			// control work first: it cannot run beside a resident launch — but the searches that are waiting need not wait for ALL of it
			// (a long upload + link of one mirror used to stall the searches of every mirror): ordinary blocking launches run beside the
			// control work's kernels, one batch at a time, until the count is down and a session can be opened again
			batch.clear();
			{
				std::unique_lock<std::mutex> lk(g_q_mu);
				if (!g_q.empty()) take_batch(batch);
				g_q_waiting.store((long) g_q.size(), std::memory_order_release);
			}
			if (!batch.empty()) run_batch(0, batch, pin);
			else std::this_thread::sleep_for(std::chrono::microseconds(50));
			continue;
		}
		batch.clear();
		EntryP e;
		size_t ef = 0, backlog = 0;
		{
			std::unique_lock<std::mutex> lk(g_q_mu);
			HGS_TIMED_WAIT(g_q_cv, lk, std::chrono::milliseconds(50), [] { return g_stop.load() || !g_q.empty(); });
			if (g_stop.load()) break;
			if (g_q.empty()) continue;
			e = g_q.front().e; ef = g_q.front().h.aux; backlog = g_q.size();
		}
		if (!e->try_read()) { std::this_thread::sleep_for(std::chrono::microseconds(50)); continue; }      // a writer is at work
		SessionP nn = open_session(e, ef, backlog);
		if (nn)
		{
			set_session(nn);
			continue;                        // (the loop above moves the queue into it)
		}
		// no stream for this shape (a beam too wide for the team form, ...): one blocking batch, as --lanes 0 does
		e->end_read();
		{
			std::unique_lock<std::mutex> lk(g_q_mu);
			if (!g_q.empty()) take_batch(batch);
			g_q_waiting.store((long) g_q.size(), std::memory_order_release);
		}
		if (!batch.empty()) run_batch(0, batch, pin);
	}
	SessionP ss = current_session();
	if (ss) close_session(ss, "shutdown", HGS_ERR_SHUTDOWN);           // (the answer threads are still at work: outstanding walks get their answers)
	g_sessions_over.store(true, std::memory_order_release);             // now they may leave
	answer_leftovers();
}

// ----------------------------------------------------------------------------- control thread
struct Mapping            // a received memfd, mapped
{
	void *p = nullptr; size_t bytes = 0; int fd = -1;
	~Mapping()
	{
		if (p) munmap(p, bytes);
		if (fd >= 0) close(fd);
	}
	bool map(int f, bool writable)
	{
		fd = f;
		struct stat st;
		if (f < 0 || fstat(f, &st) != 0 || st.st_size <= 0) return false;
		// only a memfd sealed against shrinking: a client that truncates the file under our mapping
		// would otherwise take the server down with SIGBUS
		const int seals = fcntl(f, F_GET_SEALS);
		if (seals < 0 || !(seals & F_SEAL_SHRINK)) return false;
		void *m = mmap(nullptr, (size_t) st.st_size, writable ? PROT_READ | PROT_WRITE : PROT_READ, MAP_SHARED, f, 0);
		if (m == MAP_FAILED) return false;
		p = m; bytes = (size_t) st.st_size;
		return true;
	}
};

bool meta_sane(const HnswMetadata &m)
{
	if (m.dim == 0 || m.dim > (1u << 20) || m.maxM == 0 || m.maxM > 4096) return false;
	if (m.offset_data != (m.maxM + 1) * 4 || m.offset_label != m.offset_data + m.dim * 4) return false;
	if (m.size_data_per_element != m.offset_label + 8) return false;            // embedding.c:225-228
	if ((unsigned) m.dist_func > 2) return false;
	return true;
}

void do_upload(CReq &r)
{
	HnswMetadata meta;
	if (r.payload.size() != sizeof(meta)) { r.c->respond(r.h, HGS_ERR_PROTOCOL); return; }
	memcpy(&meta, r.payload.data(), sizeof(meta));
	const size_t n = (size_t) r.h.a0;
	if (!meta_sane(meta) || n > 0xFFFFFFFEull) { r.c->respond(r.h, HNSW_GPU_ERR_ARG); return; }
	Mapping m;
	const void *elements = nullptr;
	if (n > 0)
	{
		int fd = r.fd; r.fd = -1;
		if (!m.map(fd, false) || m.bytes / meta.size_data_per_element < n) { r.c->respond(r.h, HGS_ERR_PROTOCOL); return; }
		elements = m.p;
	}
	// a1 = 1 + the content version the uploader saw at its LOOKUP (0 = unconditional): refuse a snapshot whose
	// walk raced with a change of the mirror (an insert's BIND, another backend's upload, a VACUUM's DROP)
	auto guard_ok = [&r]() {
		if (r.h.a1 == 0) return true;
		EntryP cur = find_entry(r.h.key);
		return (cur ? cur->version.load() : g_absent_ver.load()) == r.h.a1 - 1;
	};
	if (!guard_ok())
	{
		EntryP cur = find_entry(r.h.key);
		r.c->respond(r.h, HGS_ERR_STALE, cur ? cur->count.load() : 0, 0, nullptr, 0, nullptr, 0, cur ? cur->gen.load() : 0);
		return;
	}
	EntryP e = std::make_shared<Entry>();
	e->key = r.h.key; e->gen.store(r.h.gen); e->meta = meta;
	e->version.store(++g_vseq);
	int rc;
	bool dropped_own = false;
	while (true)
	{
		rc = hnsw_gpu_index_create_from_flat(&meta, elements, n, g_opt.device, &e->ix);
		if (rc == HNSW_GPU_OK || (rc != HNSW_GPU_ERR_HIP && rc != HNSW_GPU_ERR_NOMEM)) break;
		// out of device memory?  first the generation this upload replaces, then idle mirrors, LRU first
		if (!dropped_own)
		{
			dropped_own = true;
			bool had;
			{
				std::lock_guard<std::mutex> lk(g_map_mu);
				had = g_map.erase(r.h.key) > 0;
			}
			if (had) { note_removed(); continue; }
		}
		if (!evict_one(r.h.key)) break;
	}
	if (rc != HNSW_GPU_OK)
	{
		logf("upload of key %llx (%zu elements) failed: %s", (unsigned long long) r.h.key, n, hnsw_gpu_last_error());
		r.c->respond(r.h, rc);
		return;
	}
	e->count.store(n);
	e->last_used.store(now_ns());
	{
		// (control requests are handled by one thread, so nothing changed the map since the guard was checked —
		// except an eviction of this very key by the retry loop above, which the uploader asked for)
		std::lock_guard<std::mutex> lk(g_map_mu);
		g_map[r.h.key] = e;          // an older generation is freed when its last batch lets go
	}
	g_cnt.uploads++;
	g_cnt.upload_bytes += n * meta.size_data_per_element;
	VLOG("mirror %llx gen %llu: %zu elements x %zu dims", (unsigned long long) r.h.key, (unsigned long long) r.h.gen, n,
		 (size_t) meta.dim);
	r.c->respond(r.h, HGS_OK, n);
}

struct WriteLock
{
	Entry *e;
	explicit WriteLock(Entry *e_) : e(e_) { e->begin_write(); }
	~WriteLock() { e->end_write(); }
};

void do_update(CReq &r)
{
	EntryP e = find_entry(r.h.key);
	if (!e) { r.c->respond(r.h, HGS_ERR_NOKEY); return; }
	uint64_t expect = 0;
	if (r.payload.size() != 8) { r.c->respond(r.h, HGS_ERR_PROTOCOL); return; }
	memcpy(&expect, r.payload.data(), 8);
	if (expect != e->gen.load()) { r.c->respond(r.h, HGS_ERR_STALE, 0, 0, nullptr, 0, nullptr, 0, e->gen.load()); return; }
	const size_t first = (size_t) r.h.a0, count = (size_t) r.h.a1, esz = e->meta.size_data_per_element;
	Mapping m;
	int fd = r.fd; r.fd = -1;
	if (count == 0 || !m.map(fd, false) || m.bytes / esz < count) { r.c->respond(r.h, HGS_ERR_PROTOCOL); return; }
	int rc;
	{
		WriteLock wl(e.get());
		rc = hnsw_gpu_index_update_from_flat(e->ix, m.p, first, count);
		if (rc == HNSW_GPU_OK) { e->count.store(hnsw_gpu_index_count(e->ix)); e->gen.store(r.h.gen); }
		e->version.store(++g_vseq);
	}
	if (rc != HNSW_GPU_OK) logf("update of key %llx failed: %s", (unsigned long long) r.h.key, hnsw_gpu_last_error());
	g_cnt.updates++;
	r.c->respond(r.h, rc, e->count.load());
}

// hnsw_bind_point on the server's mirror: same steps as the in-process shim (embedding_shim.cpp).
void do_bind(CReq &r)
{
	EntryP e = find_entry(r.h.key);
	if (!e) { r.c->respond(r.h, HGS_ERR_NOKEY); return; }
	if (r.h.gen && r.h.gen != e->gen.load()) { r.c->respond(r.h, HGS_ERR_STALE, 0, 0, nullptr, 0, nullptr, 0, e->gen.load()); return; }
	const size_t dim = e->meta.dim, maxM = e->meta.maxM;
	const idx_t idx = r.h.aux;
	if (r.payload.size() != dim * 4) { r.c->respond(r.h, HGS_ERR_PROTOCOL); return; }
	const coord_t *point = (const coord_t *) r.payload.data();
	std::vector<uint32_t> out;
	std::vector<idx_t> mine(maxM + 1), others(maxM * (maxM + 1));
	bool fused = false;
	int rc = HNSW_GPU_OK;
	{
		WriteLock wl(e.get());
		size_t have = hnsw_gpu_index_count(e->ix);
		const size_t max_gap = std::max<size_t>(4096, 2 * (size_t) e->meta.elems_per_page);
		if (have < (size_t) idx && (size_t) idx - have > max_gap)
			rc = HNSW_GPU_ERR_ARG;                // not a page-tail hole: the mirror is not this index's
		else if (have < (size_t) idx)         // page-tail holes (embedding.c:229,693): dead placeholders
		{
			const size_t gap = (size_t) idx - have;
			std::vector<coord_t> zeros(gap * dim, 0.f);
			std::vector<label_t> dead(gap, (label_t) 1 << HNSW_LABEL_DELETED_BIT);
			// (room is asked for geometrically and only when needed: a reserve reallocates and copies the whole mirror)
			if ((size_t) idx + 1 > hnsw_gpu_index_capacity(e->ix)) rc = hnsw_gpu_index_reserve(e->ix, (size_t) idx + 1 + (size_t) idx / 2);
			if (rc == HNSW_GPU_OK) rc = hnsw_gpu_index_append(e->ix, zeros.data(), dead.data(), gap);
			have = (size_t) idx;
		}
		if (rc == HNSW_GPU_OK && have == (size_t) idx)
		{
			// the row, its links and the changed lists in one call with no host wait between the steps (hnsw_gpu_index_insert_one)
			label_t label = (label_t) r.h.a0;
			if ((size_t) idx + 1 > hnsw_gpu_index_capacity(e->ix)) rc = hnsw_gpu_index_reserve(e->ix, (size_t) idx + 1 + (size_t) idx / 2);
			if (rc == HNSW_GPU_OK) rc = hnsw_gpu_index_insert_one(e->ix, point, label, idx, mine.data(), others.data());
			fused = rc == HNSW_GPU_OK;
		}
		else if (rc == HNSW_GPU_OK && have == (size_t) idx + 1)
		{
			// the mirror came from a walk that ran after the host had stored the row: there the new element is
			// still unlinked, hence unreachable, hence a zero placeholder (host_walk.h) — give it its row and label
			std::vector<unsigned char> img(e->meta.size_data_per_element, 0);
			memcpy(img.data() + e->meta.offset_data, point, dim * 4);
			const label_t label = (label_t) r.h.a0;
			memcpy(img.data() + e->meta.offset_label, &label, sizeof(label));
			rc = hnsw_gpu_index_update_from_flat(e->ix, img.data(), idx, 1);
		}
		else if (rc == HNSW_GPU_OK)
			rc = HNSW_GPU_ERR_ARG;
		out.push_back(0);
		if (rc == HNSW_GPU_OK && idx != 0)       // bindPoint: nothing to do for the first element, hnswalg.cpp:228
		{
			if (!fused)                           // the mirror already held the row: link it, then one launch for all changed lists
			{
				rc = hnsw_gpu_index_link(e->ix, idx, 1, 1, 0, nullptr);
				if (rc == HNSW_GPU_OK) rc = hnsw_gpu_index_get_link_lists(e->ix, idx, mine.data(), others.data());
			}
			for (uint32_t j = 0; rc == HNSW_GPU_OK && j < mine[0]; j++)
			{
				out.push_back(mine[1 + j]);
				out.insert(out.end(), others.begin() + (size_t) j * (maxM + 1), others.begin() + (size_t) (j + 1) * (maxM + 1));
				out[0]++;
			}
			out.push_back(idx);
			out.insert(out.end(), mine.begin(), mine.end());
			out[0]++;
		}
		e->count.store(hnsw_gpu_index_count(e->ix));
		if (rc == HNSW_GPU_OK && r.h.a1) e->gen.store(r.h.a1);
		e->version.store(++g_vseq);
	}
	g_cnt.binds++;
	if (rc != HNSW_GPU_OK)
	{
		logf("bind of element %u into key %llx failed: %s", (unsigned) idx, (unsigned long long) r.h.key, hnsw_gpu_last_error());
		r.c->respond(r.h, rc);
		return;
	}
	r.c->respond(r.h, HGS_OK, e->count.load(), 0, out.data(), out.size() * 4, nullptr, 0, e->gen.load());
}

void do_control(CReq &r)
{
	// (everything here may free or allocate device memory, launch kernels or wait for the device; a dropped mirror dies where its
	// last reference goes, which is inside this function)
	// ... but only the operations that do: SETGEN and a malformed request touch no device, and a trickle of them must not tear a
	// resident session down again and again
	std::unique_ptr<DeviceWanted> wanted;
	switch (r.h.op)
	{
	case HGS_OP_UPLOAD: case HGS_OP_UPDATE: case HGS_OP_BIND: case HGS_OP_DROP: case HGS_OP_LINK: case HGS_OP_EXPORT: case HGS_OP_SET_DELETED:
	case HGS_OP_DIST: case HGS_OP_SHARD_ATTACH: case HGS_OP_SHARD_SEARCH:
		wanted.reset(new DeviceWanted());
		break;
	default: break;
	}
	switch (r.h.op)
	{
	case HGS_OP_UPLOAD: do_upload(r); break;
	case HGS_OP_UPDATE: do_update(r); break;
	case HGS_OP_BIND:   do_bind(r); break;
	case HGS_OP_DROP:
	{
		EntryP e;
		{
			std::lock_guard<std::mutex> lk(g_map_mu);
			auto it = g_map.find(r.h.key);
			if (it != g_map.end()) { e = it->second; g_map.erase(it); }
		}
		note_removed();                                    // also when there was nothing to drop: the caller invalidated something
		r.c->respond(r.h, e ? HGS_OK : HGS_ERR_NOKEY);
		break;
	}
	case HGS_OP_LINK:
	{
		EntryP e = find_entry(r.h.key);
		if (!e) { r.c->respond(r.h, HGS_ERR_NOKEY); break; }
		int rc;
		{
			WriteLock wl(e.get());
			rc = hnsw_gpu_index_link(e->ix, (size_t) r.h.a0, (size_t) r.h.a1, r.h.aux, 0, nullptr);
			e->version.store(++g_vseq);
			idx_t probe[4097];
			if (rc == HNSW_GPU_OK && hnsw_gpu_index_count(e->ix) > 0)
				rc = hnsw_gpu_index_get_links(e->ix, 0, probe);          // waits for the build
		}
		if (rc != HNSW_GPU_OK) logf("link failed: %s", hnsw_gpu_last_error());
		r.c->respond(r.h, rc);
		break;
	}
	case HGS_OP_EXPORT:
	{
		EntryP e = find_entry(r.h.key);
		int fd = r.fd; r.fd = -1;
		Mapping m;
		if (!e) { if (fd >= 0) close(fd); r.c->respond(r.h, HGS_ERR_NOKEY); break; }
		const size_t need = e->count.load() * e->meta.size_data_per_element;
		if (need == 0) { if (fd >= 0) close(fd); r.c->respond(r.h, HGS_OK, 0); break; }
		if (!m.map(fd, true) || m.bytes < need) { r.c->respond(r.h, HGS_ERR_PROTOCOL); break; }
		int rc;
		{
			WriteLock wl(e.get());
			rc = hnsw_gpu_index_export_flat(e->ix, m.p);
		}
		r.c->respond(r.h, rc, e->count.load());
		break;
	}
	case HGS_OP_SETGEN:
	{
		EntryP e = find_entry(r.h.key);
		uint64_t expect = 0;
		if (!e) { r.c->respond(r.h, HGS_ERR_NOKEY); break; }
		if (r.payload.size() != 8) { r.c->respond(r.h, HGS_ERR_PROTOCOL); break; }
		memcpy(&expect, r.payload.data(), 8);
		if (!e->gen.compare_exchange_strong(expect, r.h.gen))
		{
			r.c->respond(r.h, HGS_ERR_STALE, 0, 0, nullptr, 0, nullptr, 0, expect);
			break;
		}
		e->version.store(++g_vseq);
		r.c->respond(r.h, HGS_OK, e->count.load());
		break;
	}
	case HGS_OP_SET_DELETED:
	{
		EntryP e = find_entry(r.h.key);
		if (!e) { r.c->respond(r.h, HGS_ERR_NOKEY); break; }
		int rc;
		{
			WriteLock wl(e.get());
			rc = hnsw_gpu_index_set_deleted(e->ix, r.h.aux, r.h.a0 ? 1 : 0);
			e->version.store(++g_vseq);
		}
		r.c->respond(r.h, rc);
		break;
	}
	case HGS_OP_SHARD_ATTACH:
	{
		// a front hands over its exchange buffer: map it for this connection (csrc/gpu_sharded.hip, hnsw_gpu_shared_open)
		if (r.payload.size() != sizeof(hnsw_gpu_ipc_handle) || r.h.a0 == 0) { r.c->respond(r.h, HGS_ERR_PROTOCOL); break; }
		if (r.c->shard_buf) { (void) hnsw_gpu_shared_close(g_opt.device, r.c->shard_buf); r.c->shard_buf = nullptr; r.c->shard_bytes = 0; }
		hnsw_gpu_ipc_handle hnd;
		memcpy(&hnd, r.payload.data(), sizeof(hnd));
		void *p = nullptr;
		const int rc = hnsw_gpu_shared_open(g_opt.device, &hnd, &p);
		if (rc != HNSW_GPU_OK) { logf("shard attach: %s", hnsw_gpu_last_error()); r.c->respond(r.h, rc); break; }
		r.c->shard_buf = p; r.c->shard_bytes = (size_t) r.h.a0;
		r.c->respond(r.h, HGS_OK);
		break;
	}
	case HGS_OP_SHARD_SEARCH:
	{
		// this server's shard of a front's batch: hnsw_search for every query, the lists written where the front merges them
		EntryP e = find_entry(r.h.key);
		if (!e) { r.c->respond(r.h, HGS_ERR_NOKEY); break; }
		const size_t nq = (size_t) r.h.a0, ef = r.h.aux, dim = e->meta.dim;
		const size_t off_l = (size_t) r.h.a1, off_d = (size_t) r.h.gen;
		if (!r.c->shard_buf || nq == 0 || ef == 0 || nq > g_opt.max_batch || r.payload.size() != nq * dim * 4 ||
			off_l % 8 || off_d % 4 || off_l + nq * ef * 8 > r.c->shard_bytes || off_d + nq * ef * 4 > r.c->shard_bytes)
		{
			r.c->respond(r.h, HGS_ERR_PROTOCOL);
			break;
		}
		static thread_local Pinned pin;                      // queries + counts where the kernel reads / writes them directly
		const size_t qb = (nq * dim * 4 + 255) & ~(size_t) 255;
		if (!pin.reserve(qb + nq * 4)) { r.c->respond(r.h, HNSW_GPU_ERR_NOMEM); break; }
		memcpy(pin.p, r.payload.data(), nq * dim * 4);
		int rc;
		float ms = 0.f;
		e->begin_read();
		rc = hnsw_gpu_search_batch_dev(e->ix, (const coord_t *) pin.p, nq, ef, (label_t *) ((char *) r.c->shard_buf + off_l),
									   (dist_t *) ((char *) r.c->shard_buf + off_d), (uint32_t *) ((char *) pin.p + qb), nullptr, nullptr);
		if (rc == HNSW_GPU_OK) rc = hnsw_gpu_last_search_ms(e->ix, &ms);      // (waits for the launch: the lists are in the front's memory)
		e->end_read();
		e->last_used.store(now_ns());
		g_cnt.batches++;
		g_cnt.searches += nq;
		g_cnt.kernel_ns += (uint64_t) (ms * 1e6f);
		if (rc != HNSW_GPU_OK) { g_cnt.search_errors += nq; logf("shard search of %zu (ef %zu): %s", nq, ef, hnsw_gpu_last_error()); }
		r.c->respond(r.h, rc, nq);
		break;
	}
	case HGS_OP_DIST:
	{
		const size_t dim = (size_t) r.h.a0;
		if (dim == 0 || r.payload.size() != 2 * dim * 4 || r.h.aux > 2) { r.c->respond(r.h, HGS_ERR_PROTOCOL); break; }
		const coord_t *a = (const coord_t *) r.payload.data();
		dist_t out = 0;
		int rc = hnsw_gpu_dist_batch((dist_func_t) r.h.aux, a, a + dim, 1, dim, &out);
		r.c->respond(r.h, rc, 0, 0, &out, sizeof(out));
		break;
	}
	default:
		r.c->respond(r.h, HGS_ERR_PROTOCOL);
	}
	if (r.fd >= 0) { close(r.fd); r.fd = -1; }
}

void control_main()
{
	pthread_setname_np(pthread_self(), "hgs-control");
	while (true)
	{
		CReq r;
		{
			std::unique_lock<std::mutex> lk(g_c_mu);
			g_c_cv.wait(lk, [] { return g_stop.load() || !g_c.empty(); });
			if (g_stop.load()) break;
			r = std::move(g_c.front());
			g_c.pop_front();
		}
		try { do_control(r); }
		catch (const std::exception &ex)      // an allocation that failed must cost one request, not every mirror
		{
			logf("control request %u failed: %s", (unsigned) r.h.op, ex.what());
			if (r.fd >= 0) { close(r.fd); r.fd = -1; }
			r.c->respond(r.h, HNSW_GPU_ERR_NOMEM);
		}
	}
	std::lock_guard<std::mutex> lk(g_c_mu);
	for (CReq &r : g_c)
	{
		if (r.fd >= 0) close(r.fd);
		r.c->respond(r.h, HGS_ERR_SHUTDOWN);
	}
	g_c.clear();
}

// ----------------------------------------------------------------------------- readers
void fill_stats(hgs_stats *s)
{
	memset(s, 0, sizeof(*s));
	s->connections = g_cnt.connections; s->connections_now = g_cnt.connections_now;
	s->searches = g_cnt.searches; s->batches = g_cnt.batches; s->max_batch = g_cnt.max_batch;
	s->search_errors = g_cnt.search_errors;
	s->uploads = g_cnt.uploads; s->upload_bytes = g_cnt.upload_bytes; s->updates = g_cnt.updates;
	s->binds = g_cnt.binds; s->evictions = g_cnt.evictions; s->batch_ns = g_cnt.batch_ns; s->kernel_ns = g_cnt.kernel_ns;
	s->queue_ns = g_cnt.queue_ns; s->walk_ns = g_cnt.walk_ns; s->answer_ns = g_cnt.answer_ns;
	s->shm_searches = g_cnt.shm_searches;
	{
		std::lock_guard<std::mutex> lk(g_map_mu);
		s->mirrors = g_map.size();
		for (auto &kv : g_map) s->mirror_elements += kv.second->count.load();
	}
	s->uptime_ns = (uint64_t) std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - g_t0).count();
}

bool needs_fd(uint16_t op, const hgs_hdr &h)
{
	return (op == HGS_OP_UPLOAD && h.a0 > 0) || op == HGS_OP_UPDATE || op == HGS_OP_EXPORT;
}

// One SEARCH — from the socket (handle_message) or from a backend's mailbox (shm_poller_main; the header's magic says which, and
// Conn::respond answers accordingly).  False = protocol violation, drop the connection.
bool handle_search(const ConnP &c, const hgs_hdr &h, const char *payload)
{
	if (c->inflight.fetch_add(1) >= MAX_INFLIGHT_PER_CONN) { c->respond(h, HGS_ERR_PROTOCOL); return false; }
	EntryP e = find_entry(h.key);
	if (!e) { c->respond(h, HGS_ERR_NOKEY); return true; }
	const uint64_t gen = e->gen.load();
	if (h.gen && h.gen != gen) { c->respond(h, HGS_ERR_STALE, 0, 0, nullptr, 0, nullptr, 0, gen); return true; }
	if (h.len != e->meta.dim * 4 || h.aux == 0) { c->respond(h, HNSW_GPU_ERR_ARG); return true; }
	SReq r;
	r.c = c; r.e = std::move(e); r.h = h;
	r.q.resize(h.len / 4);
	memcpy(r.q.data(), payload, h.len);
	r.t_in = now_ns();
	if (g_opt.stream)
	{
		// straight into the ring of the open session when it serves this (mirror, beam).  (Not "unless older requests are still
		// queued": a backend has one search outstanding, so there is no order between requests to keep — and that rule turned one
		// queued request into a convoy: everything behind it went through the manager thread, 0.35-0.4 M q/s, profiles/r4l_*.)
		SessionP ss = current_session();
		if (ss && ss->e.get() == r.e.get() && ss->ef == h.aux && ss->submit(r)) return true;
	}
	{
		std::lock_guard<std::mutex> lk(g_q_mu);
		g_q.push_back(std::move(r));
		g_q_waiting.store((long) g_q.size(), std::memory_order_release);
	}
	g_q_cv.notify_one();
	return true;
}

// ----------------------------------------------------------------------------- mailboxes (HGS_OP_SHM, include/hnsw_gpu_server.h)
// Poller k owns the connections registered with it: nobody else looks at their mailboxes, so "taken" needs no word in shared memory.
struct ShmSet
{
	std::mutex mu;
	std::vector<ConnP> conns;
	std::atomic<uint64_t> gen{0};
};
constexpr int MAX_SHM_POLLERS = 16;
ShmSet g_shm[MAX_SHM_POLLERS];
std::atomic<unsigned> g_shm_next{0};

void shm_unregister(const ConnP &c)
{
	for (int k = 0; k < g_opt.shm_pollers; k++)
	{
		std::lock_guard<std::mutex> lk(g_shm[k].mu);
		auto &v = g_shm[k].conns;
		const size_t before = v.size();
		v.erase(std::remove(v.begin(), v.end(), c), v.end());
		if (v.size() != before) g_shm[k].gen.fetch_add(1, std::memory_order_release);
	}
}

// HGS_OP_SHM: a0 = query capacity (floats, even), a1 = result capacity, fd = the memfd.  Once per connection.
bool handle_shm(const ConnP &c, const hgs_hdr &h)
{
	if (c->fds.empty()) { c->respond(h, HGS_ERR_PROTOCOL); return false; }
	const int fd = c->fds.front();
	c->fds.pop_front();
	struct stat st;
	const uint64_t qcap = h.a0, rcap = h.a1;
	const int seals = fcntl(fd, F_GET_SEALS);
	bool ok = g_opt.shm_pollers > 0 && !c->shm && fstat(fd, &st) == 0 && seals >= 0 && (seals & F_SEAL_SHRINK) &&
			  qcap >= 2 && qcap <= (1u << 20) && (qcap & 1u) == 0 && rcap >= 1 && rcap <= (1u << 20) &&
			  (uint64_t) st.st_size >= HGS_SHM_BYTES(qcap, rcap) && (uint64_t) st.st_size <= ((uint64_t) 64 << 20);
	void *m = MAP_FAILED;
	if (ok) m = mmap(nullptr, (size_t) st.st_size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
	close(fd);
	if (!ok || m == MAP_FAILED) { c->respond(h, HNSW_GPU_ERR_ARG); return true; }          // the backend stays on the socket
	c->shm = (hgs_shm *) m; c->shm_bytes = (size_t) st.st_size;
	c->shm_qcap = (uint32_t) qcap; c->shm_rcap = (uint32_t) rcap;
	__atomic_store_n(&c->shm->state, (uint32_t) HGS_SHM_IDLE, __ATOMIC_RELEASE);
	const unsigned k = g_shm_next.fetch_add(1) % (unsigned) g_opt.shm_pollers;
	{
		std::lock_guard<std::mutex> lk(g_shm[k].mu);
		g_shm[k].conns.push_back(c);
	}
	g_shm[k].gen.fetch_add(1, std::memory_order_release);
	c->respond(h, HGS_OK, qcap, rcap);
	return true;
}

void shm_poller_main(int k)
{
	pthread_setname_np(pthread_self(), "hgs-mailbox");
	std::vector<ConnP> mine;
	uint64_t seen = ~0ull;
	unsigned idle = 0;
	while (!g_stop.load())
	{
		if (g_shm[k].gen.load(std::memory_order_acquire) != seen)
		{
			std::lock_guard<std::mutex> lk(g_shm[k].mu);
			mine = g_shm[k].conns;
			seen = g_shm[k].gen.load(std::memory_order_relaxed);
		}
		bool any = false;
		for (const ConnP &c : mine)
		{
			if (c->shm_busy.load(std::memory_order_acquire) || c->closed.load(std::memory_order_relaxed)) continue;
			hgs_shm *m = c->shm;
			if (__atomic_load_n(&m->state, __ATOMIC_ACQUIRE) != (uint32_t) HGS_SHM_POSTED) continue;
			any = true;
			hgs_hdr h;
			memcpy(&h, &m->req, sizeof(h));                      // one copy: what is checked is what is used
			c->shm_busy.store(true, std::memory_order_release);
			const bool fits = h.magic == HGS_MAGIC && h.op == HGS_OP_SEARCH && h.len <= (uint64_t) c->shm_qcap * 4u && h.aux <= c->shm_rcap;
			h.magic = HGS_SHM_MAGIC;                             // (the answer goes back through the mailbox)
			h.op = HGS_OP_SEARCH;
			if (!fits) { c->inflight++; c->respond(h, HGS_ERR_PROTOCOL); continue; }
			g_cnt.shm_searches++;
			bool keep = false;
			try { keep = handle_search(c, h, reinterpret_cast<const char *>(m) + HGS_SHM_DATA); }
			catch (const std::exception &ex) { logf("mailbox request failed: %s", ex.what()); }
			if (!keep) { c->closed.store(true); shutdown(c->fd, SHUT_RDWR); }      // (its reader thread takes the connection apart)
		}
		// nothing posted: look again at once for a while (a search is a fraction of a millisecond away), then nap
		if (any) idle = 0;
		else if (++idle < 4096) __builtin_ia32_pause();
		else std::this_thread::sleep_for(std::chrono::microseconds(idle < 8192 ? 20 : 100));
	}
}

// One complete request.  False = protocol violation, drop the connection.
bool handle_message(const ConnP &c, const hgs_hdr &h, const char *payload)
{
	switch (h.op)
	{
	case HGS_OP_HELLO:
		if (h.a0 != HGS_VERSION) { c->respond(h, HGS_ERR_PROTOCOL, HGS_VERSION); return false; }
		c->respond(h, HGS_OK, HGS_VERSION, (uint64_t) hnsw_gpu_device_count());
		return true;
	case HGS_OP_LOOKUP:
	{
		EntryP e = find_entry(h.key);
		const uint64_t ver = e ? e->version.load() : g_absent_ver.load();   // payload: the content version an UPLOAD may be guarded by
		if (e) c->respond(h, HGS_OK, e->count.load(), 1, &ver, sizeof(ver), nullptr, 0, e->gen.load());
		else c->respond(h, HGS_OK, 0, 0, &ver, sizeof(ver));
		return true;
	}
	case HGS_OP_STATS:
	{
		hgs_stats s;
		fill_stats(&s);
		c->respond(h, HGS_OK, 0, 0, &s, sizeof(s));
		return true;
	}
	case HGS_OP_SEARCH:
		return handle_search(c, h, payload);
	case HGS_OP_SHM:
		return handle_shm(c, h);
	case HGS_OP_UPLOAD: case HGS_OP_UPDATE: case HGS_OP_BIND: case HGS_OP_DROP: case HGS_OP_LINK:
	case HGS_OP_EXPORT: case HGS_OP_SET_DELETED: case HGS_OP_DIST: case HGS_OP_SETGEN:
	case HGS_OP_SHARD_ATTACH: case HGS_OP_SHARD_SEARCH:
	{
		CReq r;
		r.c = c; r.h = h;
		r.payload.assign(payload, payload + h.len);
		if (needs_fd(h.op, h))
		{
			if (c->fds.empty()) { c->respond(h, HGS_ERR_PROTOCOL); return false; }
			r.fd = c->fds.front();
			c->fds.pop_front();
		}
		{
			std::lock_guard<std::mutex> lk(g_c_mu);
			g_c.push_back(std::move(r));
		}
		g_c_cv.notify_one();
		return true;
	}
	default:
		c->respond(h, HGS_ERR_PROTOCOL);
		return false;
	}
}

// Drain the socket; false when the connection is finished.
bool on_readable(const ConnP &c)
{
	while (true)
	{
		char buf[65536];
		struct iovec iov = { buf, sizeof(buf) };
		struct msghdr mh;
		memset(&mh, 0, sizeof(mh));
		mh.msg_iov = &iov; mh.msg_iovlen = 1;
		alignas(struct cmsghdr) char cbuf[CMSG_SPACE(8 * sizeof(int))];
		mh.msg_control = cbuf; mh.msg_controllen = sizeof(cbuf);
		ssize_t n = recvmsg(c->fd, &mh, MSG_CMSG_CLOEXEC | MSG_DONTWAIT);
		if (n < 0)
		{
			if (errno == EINTR) continue;
			if (errno == EAGAIN || errno == EWOULDBLOCK) break;
			return false;
		}
		if (n == 0) return false;
		for (struct cmsghdr *cm = CMSG_FIRSTHDR(&mh); cm; cm = CMSG_NXTHDR(&mh, cm))
			if (cm->cmsg_level == SOL_SOCKET && cm->cmsg_type == SCM_RIGHTS)
			{
				size_t cnt = (cm->cmsg_len - CMSG_LEN(0)) / sizeof(int);
				for (size_t i = 0; i < cnt; i++)
				{
					int f;
					memcpy(&f, CMSG_DATA(cm) + i * sizeof(int), sizeof(int));
					if (c->fds.size() < 4) c->fds.push_back(f); else close(f);
				}
			}
		c->in.insert(c->in.end(), buf, buf + n);
		size_t off = 0;
		while (c->in.size() - off >= sizeof(hgs_hdr))
		{
			hgs_hdr h;
			memcpy(&h, c->in.data() + off, sizeof(h));
			if (h.magic != HGS_MAGIC || h.len > HGS_MAX_PAYLOAD) return false;
			if (c->in.size() - off < sizeof(h) + h.len) break;
			bool keep;
			try { keep = handle_message(c, h, c->in.data() + off + sizeof(h)); }
			catch (const std::exception &ex)     // e.g. bad_alloc while queueing a payload: this connection only
			{
				logf("request %u failed: %s", (unsigned) h.op, ex.what());
				return false;
			}
			if (!keep) return false;
			off += sizeof(h) + h.len;
		}
		if (off) c->in.erase(c->in.begin(), c->in.begin() + (long) off);
		if ((size_t) n < sizeof(buf)) break;
	}
	return !c->closed.load();
}

void reader_main(int epfd)
{
	pthread_setname_np(pthread_self(), "hgs-reader");
	std::vector<struct epoll_event> evs(256);
	while (!g_stop.load())
	{
		int n = epoll_wait(epfd, evs.data(), (int) evs.size(), -1);
		if (n < 0) { if (errno == EINTR) continue; break; }
		for (int i = 0; i < n; i++)
		{
			if (evs[i].data.ptr == nullptr) continue;            // the wake-up eventfd
			ConnP *holder = (ConnP *) evs[i].data.ptr;
			bool keep = !(evs[i].events & EPOLLERR);
			if (keep && (evs[i].events & (EPOLLIN | EPOLLHUP))) keep = on_readable(*holder);
			if (!keep)
			{
				epoll_ctl(epfd, EPOLL_CTL_DEL, (*holder)->fd, nullptr);
				(*holder)->closed.store(true);
				if ((*holder)->shm) shm_unregister(*holder);
				shutdown((*holder)->fd, SHUT_RDWR);
				g_cnt.connections_now--;
				delete holder;               // the descriptor closes when the last pending request lets go
			}
		}
	}
}

void on_signal(int)
{
	g_stop.store(true);
	uint64_t one = 1;
	if (g_wake_fd >= 0) { ssize_t w = write(g_wake_fd, &one, sizeof(one)); (void) w; }
}

void usage()
{
	fprintf(stderr,
			"usage: hnsw_gpu_server --socket PATH [--device N] [--dispatchers N] [--readers N]\n"
			"                       [--max-batch N] [--lanes N] [--linger-us N --min-batch N] [--walkers auto|0..8] [--stream 0|1 --ring N]\n"
			"                       [--shm-pollers N] [--shard-peers SOCKET[,SOCKET...]] [--verbose] [--ready-fd N]\n"
			"  --shard-peers  this server is the FRONT of a row-sharded index: the listed servers hold the other shards of every mirror (same\n"
			"             key); a search runs on all shards and is merged here (needs --lanes 0 --stream 0)\n"
			"  --shm-pollers  threads that poll the backends' mailboxes (searches through shared memory instead of the socket); 0 = off (default)\n"
			"  --stream   1 = searches go through ONE resident launch per (mirror, efsearch), fed through a ring of N slots in pinned\n"
			"             memory (no batches, no launch per query); the --dispatchers threads answer; 0 (default) = launches on lanes\n"
			"  --walkers  walking waves per 8-wave block of a search launch: auto (default) = by the walks in flight over all lanes,\n"
			"             0 = the library's choice per launch (every walk gets a block while the launch is small), 1..8 = fixed\n");
}

}  // namespace

int main(int argc, char **argv)
{
	for (int i = 1; i < argc; i++)
	{
		std::string a = argv[i];
		auto val = [&](const char *name) -> const char * {
			if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", name); exit(2); }
			return argv[++i];
		};
		if (a == "--socket") g_opt.path = val("--socket");
		else if (a == "--device") g_opt.device = atoi(val("--device"));
		else if (a == "--dispatchers") g_opt.dispatchers = atoi(val("--dispatchers"));
		else if (a == "--readers") g_opt.readers = atoi(val("--readers"));
		else if (a == "--max-batch") g_opt.max_batch = (size_t) atol(val("--max-batch"));
		else if (a == "--lanes") g_opt.lanes = atoi(val("--lanes"));
		else if (a == "--linger-us") g_opt.linger_us = atol(val("--linger-us"));
		else if (a == "--walkers") { const std::string v = val("--walkers"); g_opt.walkers = v == "auto" ? -1 : atoi(v.c_str()); }
		else if (a == "--stream") g_opt.stream = atoi(val("--stream"));
		else if (a == "--ring") g_opt.ring = (size_t) atol(val("--ring"));
		else if (a == "--shm-pollers") g_opt.shm_pollers = atoi(val("--shm-pollers"));
		else if (a == "--shard-peers")
		{
			std::string v = val("--shard-peers");
			for (size_t at = 0; at <= v.size();)
			{
				const size_t comma = std::min(v.find(',', at), v.size());
				if (comma > at) g_opt.shard_peers.push_back(v.substr(at, comma - at));
				at = comma + 1;
			}
		}
		else if (a == "--min-batch") g_opt.min_batch = (size_t) atol(val("--min-batch"));
		else if (a == "--ready-fd") g_opt.ready_fd = atoi(val("--ready-fd"));
		else if (a == "--verbose") g_opt.verbose = true;
		else { usage(); return 2; }
	}
	if (g_opt.shm_pollers < 0 || g_opt.shm_pollers > MAX_SHM_POLLERS) { usage(); return 2; }
	if (!g_opt.shard_peers.empty() && (g_opt.lanes != 0 || g_opt.stream != 0))
	{
		logf("--shard-peers (a front of row shards) takes blocking batches: --lanes 0 --stream 0");
		return 2;
	}
	if (g_opt.path.empty() || g_opt.dispatchers < 1 || g_opt.readers < 1 || g_opt.max_batch < 1 || g_opt.lanes < 0 || g_opt.lanes > 16) { usage(); return 2; }
	// (the ring keeps a margin of 64 slots, Session::submit: a 64-slot ring would take nothing at all and every request would queue for ever —
	// what the CPU tier's first resident-launch run did; 256 is the smallest ring the server accepts)
	// (stream mode keeps one word per answer thread and one mark per producer thread — readers, mailbox pollers, the manager — in fixed tables)
	if (g_opt.stream && (g_opt.dispatchers > MAX_ANSWER_THREADS || g_opt.readers + g_opt.shm_pollers + 1 > MAX_PRODUCERS))
	{
		logf("--stream 1 takes at most %d dispatchers and %d readers + mailbox pollers", MAX_ANSWER_THREADS, MAX_PRODUCERS - 1);
		return 2;
	}
	if (g_opt.stream && (g_opt.ring < 256 || g_opt.ring > ((size_t) 1 << 20) || (g_opt.ring & (g_opt.ring - 1)))) { usage(); return 2; }
	if (g_opt.path.size() >= sizeof(((struct sockaddr_un *) nullptr)->sun_path)) { logf("socket path too long"); return 2; }

	// Every lane launches on its own HIP stream.  The runtime spreads streams over GPU_MAX_HW_QUEUES
	// hardware queues (4 by default) and launches that share a queue run one after the other: with
	// 8 lanes on 4 queues a launch waited 2.0 ms for 1.0 ms of kernel (profiles/r1k_server_backends.txt).
	// One queue per lane, plus the null stream's; must be in the environment before the first HIP call.
	{
		char hwq[16];
		snprintf(hwq, sizeof(hwq), "%d", std::max(4, g_opt.dispatchers * std::max(1, g_opt.lanes) + 2));
		setenv("GPU_MAX_HW_QUEUES", hwq, 0);
	}
	const int ndev = hnsw_gpu_device_count();
	if (ndev <= 0 || g_opt.device < 0 || g_opt.device >= ndev)
	{
		logf("no usable gfx950 device (visible: %d, asked for %d); there is no CPU path", ndev, g_opt.device);
		return 3;
	}
	g_t0 = std::chrono::steady_clock::now();
	{
		const int blk = hnsw_gpu_device_blocks(g_opt.device);
		if (blk > 0) g_device_blocks = blk;
	}

	{
		// one descriptor per backend: take what the hard limit allows
		struct rlimit rl;
		if (getrlimit(RLIMIT_NOFILE, &rl) == 0 && rl.rlim_cur < rl.rlim_max)
		{
			rl.rlim_cur = rl.rlim_max;
			(void) setrlimit(RLIMIT_NOFILE, &rl);
		}
	}
	signal(SIGPIPE, SIG_IGN);
	struct sigaction sa;
	memset(&sa, 0, sizeof(sa));
	sa.sa_handler = on_signal;
	sigaction(SIGTERM, &sa, nullptr);
	sigaction(SIGINT, &sa, nullptr);
	g_wake_fd = eventfd(0, EFD_CLOEXEC | EFD_NONBLOCK);

	int lfd = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC | SOCK_NONBLOCK, 0);
	if (lfd < 0) { logf("socket: %s", strerror(errno)); return 1; }
	struct sockaddr_un addr;
	memset(&addr, 0, sizeof(addr));
	addr.sun_family = AF_UNIX;
	strncpy(addr.sun_path, g_opt.path.c_str(), sizeof(addr.sun_path) - 1);
	unlink(g_opt.path.c_str());
	mode_t old = umask(0077);                       // the socket belongs to the server's user only
	int brc = bind(lfd, (struct sockaddr *) &addr, sizeof(addr));
	umask(old);
	if (brc != 0 || listen(lfd, g_opt.backlog) != 0) { logf("bind/listen %s: %s", g_opt.path.c_str(), strerror(errno)); return 1; }

	std::vector<int> epfds;
	std::vector<std::thread> threads;
	for (int r = 0; r < g_opt.readers; r++)
	{
		int ep = epoll_create1(EPOLL_CLOEXEC);
		struct epoll_event ev;
		memset(&ev, 0, sizeof(ev));
		ev.events = EPOLLIN; ev.data.ptr = nullptr;
		epoll_ctl(ep, EPOLL_CTL_ADD, g_wake_fd, &ev);
		epfds.push_back(ep);
		threads.emplace_back(reader_main, ep);
	}
	if (g_opt.stream)
	{
		for (int d = 0; d < g_opt.dispatchers; d++) threads.emplace_back(stream_answer_main, d, g_opt.dispatchers);
		threads.emplace_back(stream_manager_main);
	}
	else
		for (int d = 0; d < g_opt.dispatchers; d++) threads.emplace_back(dispatcher_main, d);
	threads.emplace_back(control_main);
	for (int k = 0; k < g_opt.shm_pollers; k++) threads.emplace_back(shm_poller_main, k);

	logf("listening on %s (device %d of %d, %d dispatchers x %d lanes, max batch %zu)", g_opt.path.c_str(), g_opt.device, ndev,
		 g_opt.dispatchers, g_opt.lanes, g_opt.max_batch);
	if (g_opt.ready_fd >= 0)
	{
		ssize_t w = write(g_opt.ready_fd, "READY\n", 6); (void) w;
		if (g_opt.ready_fd > 2) close(g_opt.ready_fd);
	}

	size_t next = 0;
	while (!g_stop.load())
	{
		struct pollfd pf[2] = { { lfd, POLLIN, 0 }, { g_wake_fd, POLLIN, 0 } };
		int pr = poll(pf, 2, -1);
		if (pr < 0) { if (errno == EINTR) continue; break; }
		if (pf[1].revents) break;
		while (true)
		{
			int fd = accept4(lfd, nullptr, nullptr, SOCK_CLOEXEC | SOCK_NONBLOCK);
			if (fd < 0) break;
			ConnP c = std::make_shared<Conn>();
			c->fd = fd;
			g_cnt.connections++;
			g_cnt.connections_now++;
			struct epoll_event ev;
			memset(&ev, 0, sizeof(ev));
			ev.events = EPOLLIN;
			ev.data.ptr = new ConnP(c);
			if (epoll_ctl(epfds[next % epfds.size()], EPOLL_CTL_ADD, fd, &ev) != 0)
			{
				delete (ConnP *) ev.data.ptr;
				g_cnt.connections_now--;
			}
			next++;
		}
	}

	g_stop.store(true);
	on_signal(0);
	g_q_cv.notify_all();
	g_c_cv.notify_all();
	for (std::thread &t : threads) t.join();
	close(lfd);
	unlink(g_opt.path.c_str());
	{
		std::lock_guard<std::mutex> lk(g_map_mu);
		g_map.clear();                               // frees the device mirrors
	}
	hgs_stats s;
	fill_stats(&s);
	logf("stopped: %llu searches in %llu batches (largest %llu)", (unsigned long long) s.searches,
		 (unsigned long long) s.batches, (unsigned long long) s.max_batch);
	return 0;
}
