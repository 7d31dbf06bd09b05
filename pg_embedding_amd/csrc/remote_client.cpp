// remote_client.cpp — libembedding_gpuc.so: the four symbols embedding.c links against
// (embedding.h:46-47,55-56), implemented as requests to hnsw_gpu_server (hnsw_gpu_server.h).
//
//   hnsw_search          hnswalg.cpp:256-277  -> SEARCH  (batched with the other backends' scans)
//   hnsw_bind_point      hnswalg.cpp:279-291  -> BIND    (serial device insert on the server's mirror)
//                                                + write-back of the changed lists via hnsw_begin_write
//   hnsw_dist_func       distfunc.c:171-174   -> one pair on the calling core, canonical order (host_dist.h);
//                                                PG_EMBEDDING_GPU_REMOTE_DIST=1 sends it to the server (DIST) instead
//   hnsw_init_dist_func  distfunc.c:159-169   -> reads PG_EMBEDDING_GPU_SERVER
//
// No HIP is linked and no search or insert is computed here: when the server cannot be reached those
// calls fail (false) with a message on stderr.  It imports the host's storage callbacks
// (embedding.h:44,48-53) like hnswalg.cpp does, for the index walk of an upload and the write-back
// of an insert.
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <linux/futex.h>
#include <sys/un.h>

#include "hgs_io.h"
#include "host_walk.h"
#include "host_dist.h"
#include "hnsw_gpu.h"      // error codes only: nothing of libhnsw_gpu.so is linked

namespace {

std::mutex g_mu;
std::string g_path;                       // socket path (connect() or PG_EMBEDDING_GPU_SERVER)

thread_local int   t_fd = -1;
thread_local pid_t t_pid = 0;
thread_local char  t_err[512] = "";
thread_local std::vector<char> t_resp;    // payload of the last response
// this connection's mailbox (HGS_OP_SHM, include/hnsw_gpu_server.h): SEARCH requests are posted there instead of written to the socket
struct Mailbox { hgs_shm *m = nullptr; size_t bytes = 0; uint32_t qcap = 0, rcap = 0; int state = 0; };   // state: 0 not tried, 1 ready, -1 none
thread_local Mailbox t_box;
std::atomic<bool> g_shm_refused{false};   // a server that does not know HGS_OP_SHM (it dropped the connection): never asked again

struct Attachment { HnswMetadata *meta; uint64_t key, gen; bool own = false; };   // own: a private mirror under an ephemeral key
std::vector<Attachment> g_attached;
std::vector<HnswMetadata *> g_building;   // metas between hnsw_gpu_remote_begin_build and ..._finish_build
std::atomic<uint64_t> g_ephemeral{0};

int fail(int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(t_err, sizeof(t_err), fmt, ap);
	va_end(ap);
	return code;
}

void drop_connection()
{
	if (t_fd >= 0) close(t_fd);
	t_fd = -1;
	if (t_box.m) munmap(t_box.m, t_box.bytes);
	t_box = Mailbox();                     // (a new connection sets up a new mailbox)
}

std::vector<std::string> split_paths(const std::string &list)
{
	std::vector<std::string> all;
	size_t a = 0;
	while (a <= list.size())
	{
		const size_t b = list.find(',', a);
		const std::string one = list.substr(a, b == std::string::npos ? std::string::npos : b - a);
		if (!one.empty()) all.push_back(one);
		if (b == std::string::npos) break;
		a = b + 1;
	}
	return all;
}

// One request on a connection of its own to the server at `path` (no payload either way).
int one_shot(const std::string &path, uint16_t op, uint64_t key)
{
	struct sockaddr_un addr;
	memset(&addr, 0, sizeof(addr));
	addr.sun_family = AF_UNIX;
	if (path.size() >= sizeof(addr.sun_path)) return HGS_ERR_IO;
	strncpy(addr.sun_path, path.c_str(), sizeof(addr.sun_path) - 1);
	int fd = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
	if (fd < 0) return HGS_ERR_IO;
	int rc = HGS_ERR_IO;
	hgs_hdr h, r;
	if (connect(fd, (struct sockaddr *) &addr, sizeof(addr)) == 0)
	{
		memset(&h, 0, sizeof(h));
		h.magic = HGS_MAGIC; h.op = HGS_OP_HELLO; h.a0 = HGS_VERSION;
		if (hgs::send_msg(fd, &h, nullptr, 0, nullptr, 0) == 0 && hgs::recv_exact(fd, &r, sizeof(r), nullptr) == 0 && r.status == HGS_OK)
		{
			memset(&h, 0, sizeof(h));
			h.magic = HGS_MAGIC; h.op = op; h.key = key;
			if (hgs::send_msg(fd, &h, nullptr, 0, nullptr, 0) == 0 && hgs::recv_exact(fd, &r, sizeof(r), nullptr) == 0 && r.len == 0)
				rc = r.status;
		}
	}
	close(fd);
	return rc;
}

int ensure_connected()
{
	if (t_fd >= 0 && t_pid == getpid()) return HGS_OK;
	if (t_fd >= 0) drop_connection();             // inherited across fork(): the parent's conversation
	std::string path;
	{
		std::lock_guard<std::mutex> lk(g_mu);
		if (g_path.empty())
		{
			const char *env = getenv("PG_EMBEDDING_GPU_SERVER");
			if (env && *env) g_path = env;
		}
		path = g_path;
	}
	if (path.empty())
		return fail(HGS_ERR_IO, "no server socket: set PG_EMBEDDING_GPU_SERVER or call hnsw_gpu_remote_connect()");
	// Several GPUs = several servers (one process per GPU), sockets separated by ','.  A backend stays with
	// one of them (by process id): every server mirrors the indexes its backends use — replicas, so
	// read throughput scales with the GPUs; a mirror on another server that an insert here made stale is
	// noticed by its generation at the next attach there and uploaded again (a DROP goes to all of them).
	{
		const std::vector<std::string> all = split_paths(path);
		if (all.empty()) return fail(HGS_ERR_IO, "no server socket in \"%s\"", path.c_str());
		path = all[(size_t) getpid() % all.size()];
	}
	struct sockaddr_un addr;
	memset(&addr, 0, sizeof(addr));
	addr.sun_family = AF_UNIX;
	if (path.size() >= sizeof(addr.sun_path)) return fail(HGS_ERR_IO, "socket path too long");
	strncpy(addr.sun_path, path.c_str(), sizeof(addr.sun_path) - 1);
	int fd = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
	if (fd < 0) return fail(HGS_ERR_IO, "socket(): %s", strerror(errno));
	if (connect(fd, (struct sockaddr *) &addr, sizeof(addr)) != 0)
	{
		int e = errno;
		close(fd);
		return fail(HGS_ERR_IO, "cannot reach hnsw_gpu_server at %s: %s", path.c_str(), strerror(e));
	}
	t_fd = fd;
	t_pid = getpid();
	hgs_hdr h, r;
	memset(&h, 0, sizeof(h));
	h.magic = HGS_MAGIC; h.op = HGS_OP_HELLO; h.a0 = HGS_VERSION;
	if (hgs::send_msg(t_fd, &h, nullptr, 0, nullptr, 0) != 0 || hgs::recv_exact(t_fd, &r, sizeof(r), nullptr) != 0 ||
		r.magic != HGS_MAGIC || r.status != HGS_OK || r.len != 0)
	{
		drop_connection();
		return fail(HGS_ERR_IO, "handshake with hnsw_gpu_server at %s failed", path.c_str());
	}
	return HGS_OK;
}

// One request -> one response.  The response header lands in *r, its payload in t_resp.
int rpc(hgs_hdr *h, const void *p1, size_t l1, const void *p2, size_t l2, int pass_fd, hgs_hdr *r)
{
	int rc = ensure_connected();
	if (rc != HGS_OK) return rc;
	if (pass_fd >= 0 && fcntl(pass_fd, F_ADD_SEALS, F_SEAL_SHRINK) != 0)      // see Shm::seal
		return fail(HGS_ERR_IO, "cannot seal the shared memory file: %s", strerror(errno));
	h->magic = HGS_MAGIC;
	h->len = (uint32_t) ((p1 ? l1 : 0) + (p2 ? l2 : 0));
	// A connection that was fine at the last call may have died since (the server was restarted): one
	// fresh connection, one retry — for everything that can be repeated without harm, i.e. all but BIND
	// (an insert that may or may not have been applied is the caller's to sort out: it fails).
	// A server that has stopped answering must not hang the backend for good (and an uninterruptible recv cannot
	// be cancelled): PG_EMBEDDING_GPU_TIMEOUT_MS, default 10 minutes (bulk builds and uploads are the long requests),
	// 0 = wait for ever.  A request that timed out is not retried — it may still be running.
	static const int timeout_ms = [] {
		const char *e = getenv("PG_EMBEDDING_GPU_TIMEOUT_MS");
		const long v = e ? atol(e) : 600000;
		return v <= 0 ? -1 : (int) v;
	}();
	for (int attempt = 0;; attempt++)
	{
		if (hgs::send_msg(t_fd, h, p1, l1, p2, l2, pass_fd) == 0)
		{
			errno = 0;
			if (hgs::recv_exact(t_fd, r, sizeof(*r), nullptr, timeout_ms) == 0) break;
			if (errno == ETIMEDOUT)
			{
				drop_connection();
				return fail(HGS_ERR_IO, "hnsw_gpu_server did not answer request %u within %d ms", (unsigned) h->op, timeout_ms);
			}
		}
		drop_connection();
		if (attempt == 1 || h->op == HGS_OP_BIND || ensure_connected() != HGS_OK)
			return fail(HGS_ERR_IO, "lost the connection to hnsw_gpu_server");
	}
	if (r->magic != HGS_MAGIC || r->len > HGS_MAX_PAYLOAD || r->op != h->op)
	{
		drop_connection();
		return fail(HGS_ERR_PROTOCOL, "bad response from hnsw_gpu_server");
	}
	t_resp.resize(r->len);
	if (r->len && hgs::recv_exact(t_fd, t_resp.data(), r->len, nullptr, timeout_ms) != 0)
	{
		drop_connection();
		return fail(HGS_ERR_IO, "lost the connection to hnsw_gpu_server");
	}
	if (r->status != HGS_OK)
		return fail(r->status, "hnsw_gpu_server refused request %u: status %d", (unsigned) h->op, (int) r->status);
	return HGS_OK;
}

// An anonymous shared-memory file of `bytes` bytes, mapped.  The server maps the same pages.
struct Shm
{
	int fd = -1; void *p = nullptr; size_t bytes = 0;
	~Shm()
	{
		if (p) munmap(p, bytes);
		if (fd >= 0) close(fd);
	}
	bool create(size_t nbytes)
	{
		fd = memfd_create("hnsw_gpu_elements", MFD_CLOEXEC | MFD_ALLOW_SEALING);
		if (fd < 0) return false;
		if (nbytes == 0) nbytes = 1;
		if (ftruncate(fd, (off_t) nbytes) != 0) return false;
		void *m = mmap(nullptr, nbytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
		if (m == MAP_FAILED) return false;
		p = m; bytes = nbytes;
		return true;
	}
	// The server only maps files that can no longer shrink under it (it would die of SIGBUS).
	bool seal() { return fcntl(fd, F_ADD_SEALS, F_SEAL_SHRINK) == 0; }
	void release()
	{
		if (p) munmap(p, bytes);
		if (fd >= 0) close(fd);
		p = nullptr; bytes = 0; fd = -1;
	}
	bool grow(size_t nbytes)
	{
		if (nbytes <= bytes) return true;
		size_t nb = bytes;
		while (nb < nbytes) nb *= 2;
		if (ftruncate(fd, (off_t) nb) != 0) return false;
		void *m = mremap(p, bytes, nb, MREMAP_MAYMOVE);
		if (m == MAP_FAILED) return false;
		p = m; bytes = nb;
		return true;
	}
};

// Is there a mailbox on this connection that takes a query of `dim` floats and `ef` results?  Sets one up at the first search
// when PG_EMBEDDING_GPU_SHM=1 (default: every request on the socket).
bool mailbox_ready(size_t dim, size_t ef)
{
	if (t_box.state == 0)
	{
		t_box.state = -1;
		const char *want = getenv("PG_EMBEDDING_GPU_SHM");            // opt-in (and the server must run mailbox pollers: --shm-pollers N)
		if (!(want && *want == '1') || g_shm_refused.load()) return false;
		const uint32_t qcap = (uint32_t) std::max<size_t>(2048, (dim + 1) & ~(size_t) 1), rcap = (uint32_t) std::max<size_t>(1024, ef);
		if (dim > (1u << 20) || ef > (1u << 20)) return false;
		Shm box;
		const size_t bytes = (HGS_SHM_BYTES(qcap, rcap) + 4095) & ~(size_t) 4095;
		if (!box.create(bytes)) return false;
		hgs_shm *m = (hgs_shm *) box.p;
		memset(m, 0, sizeof(*m));
		m->qcap = qcap; m->rcap = rcap;
		hgs_hdr h, r;
		memset(&h, 0, sizeof(h));
		h.op = HGS_OP_SHM; h.a0 = qcap; h.a1 = rcap;
		const int rc = rpc(&h, nullptr, 0, nullptr, 0, box.fd, &r);
		if (rc != HGS_OK)
		{
			if (rc == HGS_ERR_PROTOCOL) g_shm_refused.store(true);     // an older server: it has dropped this connection, too
			return false;                                             // (the mapping goes with `box`)
		}
		t_box.m = m; t_box.bytes = box.bytes; t_box.qcap = qcap; t_box.rcap = rcap; t_box.state = 1;
		box.p = nullptr;                                              // the mapping stays, the descriptor is not needed any more
	}
	return t_box.state == 1 && dim <= t_box.qcap && ef <= t_box.rcap && ef > 0;
}

// One SEARCH through the mailbox.  HGS_OK / a server status (t_err set) — or 1: the connection is gone, take the socket path (which
// makes a fresh connection and retries once, as for every repeatable request).
int mailbox_search(uint64_t key, uint64_t generation, const coord_t *query, size_t dim, size_t ef, label_t *labels, dist_t *dists, size_t *count)
{
	static const int timeout_ms = [] {
		const char *e = getenv("PG_EMBEDDING_GPU_TIMEOUT_MS");
		const long v = e ? atol(e) : 600000;
		return v <= 0 ? -1 : (int) v;
	}();
	hgs_shm *m = t_box.m;
	hgs_hdr h;
	memset(&h, 0, sizeof(h));
	h.magic = HGS_MAGIC; h.op = HGS_OP_SEARCH; h.len = (uint32_t) (dim * 4); h.aux = (uint32_t) ef;
	h.key = key; h.gen = generation; h.a0 = dists ? 1 : 0;
	char *data = reinterpret_cast<char *>(m) + HGS_SHM_DATA;
	memcpy(&m->req, &h, sizeof(h));
	memcpy(data, query, dim * 4);
	__atomic_store_n(&m->sleeping, 0u, __ATOMIC_RELAXED);
	__atomic_store_n(&m->state, (uint32_t) HGS_SHM_POSTED, __ATOMIC_RELEASE);
	struct timespec t0;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	bool done = false;
	for (int spin = 0; spin < 256 && !done; spin++)                    // a walk takes a fraction of a millisecond: look a few times, then sleep
	{
		done = __atomic_load_n(&m->state, __ATOMIC_ACQUIRE) == (uint32_t) HGS_SHM_DONE;
		if (!done) __builtin_ia32_pause();
	}
	while (!done)
	{
		__atomic_store_n(&m->sleeping, 1u, __ATOMIC_SEQ_CST);           // (the server stores DONE, then looks at this word)
		if (__atomic_load_n(&m->state, __ATOMIC_SEQ_CST) == (uint32_t) HGS_SHM_DONE) break;
		struct timespec nap = { 0, 100 * 1000 * 1000 };
		(void) syscall(SYS_futex, &m->state, FUTEX_WAIT, (uint32_t) HGS_SHM_POSTED, &nap, nullptr, 0);
		if (__atomic_load_n(&m->state, __ATOMIC_ACQUIRE) == (uint32_t) HGS_SHM_DONE) break;
		// still nothing: is the server there at all?  (an answer can only be late; a closed socket means it never comes)
		struct pollfd pf = { t_fd, POLLIN, 0 };
		char peek;
		if (poll(&pf, 1, 0) > 0 && ((pf.revents & (POLLHUP | POLLERR)) || ((pf.revents & POLLIN) && recv(t_fd, &peek, 1, MSG_PEEK | MSG_DONTWAIT) == 0)))
		{
			drop_connection();
			return 1;
		}
		struct timespec t1;
		clock_gettime(CLOCK_MONOTONIC, &t1);
		const long waited = (long) (t1.tv_sec - t0.tv_sec) * 1000 + (t1.tv_nsec - t0.tv_nsec) / 1000000;
		if (timeout_ms >= 0 && waited > timeout_ms)
		{
			drop_connection();
			return fail(HGS_ERR_IO, "hnsw_gpu_server did not answer request %u within %d ms", (unsigned) HGS_OP_SEARCH, timeout_ms);
		}
	}
	__atomic_store_n(&m->sleeping, 0u, __ATOMIC_RELAXED);
	hgs_hdr r;
	memcpy(&r, &m->resp, sizeof(r));
	int rc = HGS_OK;
	const size_t cnt = (size_t) r.a0;
	if (r.magic != HGS_MAGIC || r.op != HGS_OP_SEARCH) rc = fail(HGS_ERR_PROTOCOL, "bad response from hnsw_gpu_server");
	else if (r.status != HGS_OK) rc = fail(r.status, "hnsw_gpu_server refused request %u: status %d", (unsigned) HGS_OP_SEARCH, (int) r.status);
	else if (cnt > ef || r.len != cnt * (dists ? 12 : 8)) rc = fail(HGS_ERR_PROTOCOL, "bad SEARCH response");
	else
	{
		const char *res = data + (size_t) t_box.qcap * 4u;
		memcpy(labels, res, cnt * 8);
		if (dists) memcpy(dists, res + (size_t) t_box.rcap * 8u, cnt * 4);
		*count = cnt;
	}
	__atomic_store_n(&m->state, (uint32_t) HGS_SHM_IDLE, __ATOMIC_RELEASE);
	if (rc == HGS_ERR_PROTOCOL) drop_connection();
	return rc;
}

bool find_attached(HnswMetadata *meta, Attachment *out)
{
	std::lock_guard<std::mutex> lk(g_mu);
	for (const Attachment &a : g_attached)
		if (a.meta == meta) { *out = a; return true; }
	return false;
}

// Copy the host index into shared memory by the accessor the reference search uses (hnsw_begin_read,
// embedding.c:704-757), following the links from the entry point, one pin at a time — host_walk.h
// explains why the walk must not probe element numbers past the end (the real host raises ERROR there)
// and why leaving out unreachable elements changes no answer; element numbers that were not reached
// (page-tail holes of embedding.c:229,693 among them) become vacuum-flagged placeholders.
// guard: 0 = replace whatever is there; otherwise 1 + the content version seen at the LOOKUP this walk follows
// (the server refuses with HGS_ERR_STALE when the mirror was changed in between: hnsw_gpu_server.h, UPLOAD)
int walk_and_upload(HnswMetadata *meta, uint64_t key, uint64_t gen, uint64_t guard = 0)
{
	// Per-thread and reused: a host callback that leaves by longjmp (elog(ERROR), embedding.c:715)
	// must not leak the area — the next walk takes it back.
	static thread_local Shm shm;
	shm.release();
	if (!shm.create((size_t) 1 << 20)) return fail(HGS_ERR_IO, "memfd_create/mmap failed: %s", strerror(errno));
	const long walked = hostwalk::copy_reachable(meta, [](size_t bytes) -> char * { return shm.grow(bytes) ? (char *) shm.p : nullptr; });
	if (walked < 0) return fail(HGS_ERR_IO, "cannot grow the upload area");
	const size_t n = (size_t) walked;
	hgs_hdr h, r;
	memset(&h, 0, sizeof(h));
	h.op = HGS_OP_UPLOAD; h.key = key; h.gen = gen; h.a0 = n; h.a1 = guard;
	int rc = rpc(&h, meta, sizeof(*meta), nullptr, 0, n ? shm.fd : -1, &r);
	shm.release();
	return rc;
}

uint64_t ephemeral_key()
{
	// high bit set: never collides with a relfilenode-style key; unique per process and call
	return 0x8000000000000000ull | ((uint64_t) getpid() << 24) | (g_ephemeral++ & 0xFFFFFF);
}

int simple(uint16_t op, uint64_t key, uint32_t aux, uint64_t a0, uint64_t a1, hgs_hdr *r)
{
	hgs_hdr h;
	memset(&h, 0, sizeof(h));
	h.op = op; h.key = key; h.aux = aux; h.a0 = a0; h.a1 = a1;
	return rpc(&h, nullptr, 0, nullptr, 0, -1, r);
}

}  // namespace

// ---------------------------------------------------------------------------------------
// additive client calls
// ---------------------------------------------------------------------------------------
extern "C" const char *hnsw_gpu_remote_last_error(void) { return t_err; }

static int hnsw_gpu_remote_connect_impl(const char *socket_path)
{
	if (!socket_path || !*socket_path) return fail(HGS_ERR_IO, "empty socket path");
	{
		std::lock_guard<std::mutex> lk(g_mu);
		g_path = socket_path;
	}
	drop_connection();
	return ensure_connected();
}

extern "C" void hnsw_gpu_remote_disconnect(void) { drop_connection(); }

static int hnsw_gpu_remote_lookup_impl(uint64_t key, uint64_t *generation, size_t *count, int *present)
{
	hgs_hdr r;
	int rc = simple(HGS_OP_LOOKUP, key, 0, 0, 0, &r);
	if (rc != HGS_OK) return rc;
	if (generation) *generation = r.gen;
	if (count) *count = (size_t) r.a0;
	if (present) *present = (int) r.a1;
	return HGS_OK;
}

static int hnsw_gpu_remote_upload_impl(const HnswMetadata *meta, uint64_t key, uint64_t generation,
									  const void *elements, size_t n)
{
	if (!meta || (n && !elements)) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	Shm shm;
	const size_t bytes = n * meta->size_data_per_element;
	if (n)
	{
		if (!shm.create(bytes)) return fail(HGS_ERR_IO, "memfd_create/mmap failed: %s", strerror(errno));
		memcpy(shm.p, elements, bytes);
	}
	hgs_hdr h, r;
	memset(&h, 0, sizeof(h));
	h.op = HGS_OP_UPLOAD; h.key = key; h.gen = generation; h.a0 = n;
	return rpc(&h, meta, sizeof(*meta), nullptr, 0, n ? shm.fd : -1, &r);
}

static int hnsw_gpu_remote_update_impl(uint64_t key, uint64_t expected_generation, uint64_t new_generation,
									  const HnswMetadata *meta, const void *elements, size_t first, size_t count)
{
	if (!meta || !elements || count == 0) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	Shm shm;
	const size_t bytes = count * meta->size_data_per_element;
	if (!shm.create(bytes)) return fail(HGS_ERR_IO, "memfd_create/mmap failed: %s", strerror(errno));
	memcpy(shm.p, elements, bytes);
	hgs_hdr h, r;
	memset(&h, 0, sizeof(h));
	h.op = HGS_OP_UPDATE; h.key = key; h.gen = new_generation; h.a0 = first; h.a1 = count;
	return rpc(&h, &expected_generation, 8, nullptr, 0, shm.fd, &r);
}

static int hnsw_gpu_remote_search_impl(uint64_t key, uint64_t generation, const coord_t *query, size_t dim, size_t ef,
									  label_t *labels, dist_t *dists, size_t *count)
{
	if (!query || !labels || !count) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	if (ef == 0 || ef > 0xFFFFFFFFull) return fail(HNSW_GPU_ERR_ARG, "ef out of range");
	if (ensure_connected() == HGS_OK && mailbox_ready(dim, ef))
	{
		const int mrc = mailbox_search(key, generation, query, dim, ef, labels, dists, count);
		if (mrc != 1) return mrc;                  // 1 = the connection went away: below, on a fresh one
	}
	hgs_hdr h, r;
	memset(&h, 0, sizeof(h));
	h.op = HGS_OP_SEARCH; h.key = key; h.gen = generation; h.aux = (uint32_t) ef; h.a0 = dists ? 1 : 0;
	int rc = rpc(&h, query, dim * 4, nullptr, 0, -1, &r);
	if (rc != HGS_OK) return rc;
	const size_t cnt = (size_t) r.a0;
	if (cnt > ef || t_resp.size() != cnt * (dists ? 12 : 8)) return fail(HGS_ERR_PROTOCOL, "bad SEARCH response");
	memcpy(labels, t_resp.data(), cnt * 8);
	if (dists) memcpy(dists, t_resp.data() + cnt * 8, cnt * 4);
	*count = cnt;
	return HGS_OK;
}

static int hnsw_gpu_remote_link_impl(uint64_t key, size_t first, size_t count, size_t max_batch)
{
	hgs_hdr r;
	return simple(HGS_OP_LINK, key, (uint32_t) max_batch, first, count, &r);
}

static int hnsw_gpu_remote_export_impl(uint64_t key, void *elements, size_t bytes)
{
	if (!elements) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	Shm shm;
	if (!shm.create(bytes)) return fail(HGS_ERR_IO, "memfd_create/mmap failed: %s", strerror(errno));
	hgs_hdr h, r;
	memset(&h, 0, sizeof(h));
	h.op = HGS_OP_EXPORT; h.key = key;
	int rc = rpc(&h, nullptr, 0, nullptr, 0, shm.fd, &r);
	if (rc != HGS_OK) return rc;
	memcpy(elements, shm.p, bytes);
	return HGS_OK;
}

static int hnsw_gpu_remote_set_deleted_impl(uint64_t key, idx_t idx, int deleted)
{
	hgs_hdr r;
	return simple(HGS_OP_SET_DELETED, key, idx, deleted ? 1 : 0, 0, &r);
}

static int hnsw_gpu_remote_drop_impl(uint64_t key)
{
	hgs_hdr r;
	int rc = simple(HGS_OP_DROP, key, 0, 0, 0, &r);
	// With several servers every replica has to go: a mirror that outlives a VACUUM would keep answering
	// with TIDs whose rows are gone — or, after TID reuse, are other rows.
	std::string list;
	{
		std::lock_guard<std::mutex> lk(g_mu);
		list = g_path;
	}
	const std::vector<std::string> all = split_paths(list);
	if (all.size() > 1)
	{
		const std::string &mine = all[(size_t) getpid() % all.size()];
		for (const std::string &p : all)
			if (&p != &mine)
			{
				const int r2 = one_shot(p, HGS_OP_DROP, key);
				if (r2 != HGS_OK && r2 != HGS_ERR_NOKEY)
					rc = fail(HGS_ERR_IO, "could not drop mirror %llx on the server at %s (%d)", (unsigned long long) key, p.c_str(), r2);
			}
	}
	return rc;
}

static int hnsw_gpu_remote_stats_impl(hgs_stats *out)
{
	if (!out) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	hgs_hdr r;
	int rc = simple(HGS_OP_STATS, 0, 0, 0, 0, &r);
	if (rc != HGS_OK) return rc;
	if (t_resp.size() != sizeof(*out)) return fail(HGS_ERR_PROTOCOL, "bad STATS response");
	memcpy(out, t_resp.data(), sizeof(*out));
	return HGS_OK;
}

static int hnsw_gpu_remote_attach_impl(HnswMetadata *meta, uint64_t key, uint64_t generation)
{
	if (!meta) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	{
		// an aborted CREATE INDEX (elog(ERROR) between begin_build and finish_build) may have left this address
		// registered; a meta that is being attached is not being bulk-built
		std::lock_guard<std::mutex> lk(g_mu);
		for (size_t i = 0; i < g_building.size(); i++)
			if (g_building[i] == meta) { g_building[i] = g_building.back(); g_building.pop_back(); break; }
	}
	// LOOKUP -> (walk -> guarded UPLOAD): the server takes the snapshot only if nobody changed the mirror since the
	// LOOKUP.  Backends that race (an inserter between "row stored" and "mirror renamed", two scanners uploading the
	// same index) make one of them retry; after a few rounds the scan takes a private mirror instead of waiting.
	bool own = false;
	int rc = HGS_OK;
	for (int round = 0;; round++)
	{
		hgs_hdr r;
		rc = simple(HGS_OP_LOOKUP, key, 0, 0, 0, &r);
		if (rc != HGS_OK) return rc;
		if (r.a1 && r.gen == generation) break;                  // present and current
		if (r.a1 && round == 0)
		{
			// Present under another name: most often an insert in another backend is between "row stored" and
			// "mirror renamed" (hnsw_gpu_remote_advance) and the server catches up within a millisecond or two.
			// Re-walking and re-uploading the whole index (3 GB at 1M x 768) to win that race would replace the
			// very mirror the inserter is extending, so give it a moment first.
			bool caught_up = false;
			for (int i = 0; i < 5 && !caught_up; i++)
			{
				struct timespec ts = { 0, 2000000 };
				nanosleep(&ts, nullptr);
				if (simple(HGS_OP_LOOKUP, key, 0, 0, 0, &r) != HGS_OK) break;
				caught_up = r.a1 && r.gen == generation;
			}
			if (caught_up) break;
		}
		uint64_t version = 0;
		if (t_resp.size() == sizeof(version)) memcpy(&version, t_resp.data(), sizeof(version));
		if (round == 4)
		{
			key = ephemeral_key(); generation = 1; own = true;
			rc = walk_and_upload(meta, key, generation);
			if (rc != HGS_OK) return rc;
			break;
		}
		rc = walk_and_upload(meta, key, generation, version + 1);
		if (rc == HGS_OK) break;
		if (rc != HGS_ERR_STALE) return rc;
	}
	std::lock_guard<std::mutex> lk(g_mu);
	for (Attachment &a : g_attached)
		if (a.meta == meta) { a.key = key; a.gen = generation; a.own = own; return HGS_OK; }
	Attachment at; at.meta = meta; at.key = key; at.gen = generation; at.own = own;
	g_attached.push_back(at);
	return HGS_OK;
}

// CREATE INDEX offload.  Between begin_build and finish_build hnsw_bind_point(meta, ...) only reports
// success: the host has stored the row zero-linked (embedding.c:619-621,670), nothing is linked yet.
static int hnsw_gpu_remote_begin_build_impl(HnswMetadata *meta)
{
	if (!meta) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	std::lock_guard<std::mutex> lk(g_mu);
	// One build at a time per backend (CREATE INDEX is one statement): whatever an earlier build that left by
	// elog(ERROR) registered is stale — its HnswIndex was freed and the address may be handed out again.
	g_building.clear();
	g_building.push_back(meta);
	return HGS_OK;
}

// Upload the stored rows (element numbers [0, n_slots); numbers that do not exist — page-tail holes —
// become vacuum-flagged placeholders), link them all on the device (hnsw_gpu_index_link: max_batch 1 =
// the reference's serial order, bit-identical graph; 0 = batched bulk build, an equally good graph in a
// fraction of the time), and write every element's link list back into the host's pages through
// hnsw_begin_write/hnsw_end_write.  The caller holds the index-wide writer lock (embedding.c:624-629),
// as for any hnsw_bind_point.  The mirror stays on the server as (key, generation).
static int hnsw_gpu_remote_finish_build_impl(HnswMetadata *meta, uint64_t key, uint64_t generation, size_t n_slots,
											size_t max_batch)
{
	if (!meta) return fail(HNSW_GPU_ERR_ARG, "NULL argument");
	{
		std::lock_guard<std::mutex> lk(g_mu);
		for (size_t i = 0; i < g_building.size(); i++)
			if (g_building[i] == meta) { g_building[i] = g_building.back(); g_building.pop_back(); break; }
	}
	const size_t esz = meta->size_data_per_element, lbytes = (meta->maxM + 1) * sizeof(idx_t);
	const label_t dead = (label_t) 1 << HNSW_LABEL_DELETED_BIT;
	static thread_local Shm shm;                   // reused: a callback may leave by longjmp (see walk_and_upload)
	static thread_local std::vector<uint64_t> real;
	shm.release();
	real.assign((n_slots + 63) / 64, 0);
	if (!shm.create(n_slots ? n_slots * esz : 1)) return fail(HGS_ERR_IO, "memfd_create/mmap failed: %s", strerror(errno));
	for (size_t idx = 0; idx < n_slots; idx++)
	{
		char *dst = (char *) shm.p + idx * esz;
		idx_t *links = nullptr;
		if (hnsw_begin_read(meta, (idx_t) idx, &links, nullptr, nullptr))
		{
			memcpy(dst, links, esz);
			hnsw_end_read(meta);
			memset(dst, 0, lbytes);                  // whatever the pages hold, the build starts un-linked
			real[idx / 64] |= 1ull << (idx % 64);
		}
		else
		{
			memset(dst, 0, esz);
			memcpy(dst + meta->offset_label, &dead, sizeof(dead));
		}
	}
	hgs_hdr h, r;
	memset(&h, 0, sizeof(h));
	h.op = HGS_OP_UPLOAD; h.key = key; h.gen = generation; h.a0 = n_slots;
	int rc = rpc(&h, meta, sizeof(*meta), nullptr, 0, n_slots ? shm.fd : -1, &r);
	// link the stored elements run by run: a placeholder must stay what it is in the host, absent
	for (size_t first = 0; rc == HGS_OK && first < n_slots;)
	{
		if (!((real[first / 64] >> (first % 64)) & 1)) { first++; continue; }
		size_t end = first;
		while (end < n_slots && ((real[end / 64] >> (end % 64)) & 1)) end++;
		rc = hnsw_gpu_remote_link(key, first, end - first, max_batch);
		first = end;
	}
	shm.release();
	if (rc != HGS_OK || n_slots == 0) return rc;
	// the graph comes back as element images; only the link lists go into the pages
	if (!shm.create(n_slots * esz)) return fail(HGS_ERR_IO, "memfd_create/mmap failed: %s", strerror(errno));
	memset(&h, 0, sizeof(h));
	h.op = HGS_OP_EXPORT; h.key = key;
	rc = rpc(&h, nullptr, 0, nullptr, 0, shm.fd, &r);
	for (size_t idx = 0; rc == HGS_OK && idx < n_slots; idx++)
	{
		if (!((real[idx / 64] >> (idx % 64)) & 1)) continue;
		idx_t *dst = nullptr;
		hnsw_begin_write(meta, (idx_t) idx, &dst, nullptr, nullptr);
		memcpy(dst, (char *) shm.p + idx * esz, lbytes);
		hnsw_end_write(meta);
	}
	shm.release();
	return rc;
}

static int hnsw_gpu_remote_advance_impl(HnswMetadata *meta, uint64_t new_generation)
{
	Attachment at;
	if (!meta || !find_attached(meta, &at)) return fail(HNSW_GPU_ERR_ARG, "meta is not attached");
	if (at.gen == new_generation) return HGS_OK;
	hgs_hdr h, r;
	memset(&h, 0, sizeof(h));
	h.op = HGS_OP_SETGEN; h.key = at.key; h.gen = new_generation;
	int rc = rpc(&h, &at.gen, 8, nullptr, 0, -1, &r);
	if (rc != HGS_OK && rc != HGS_ERR_STALE) return rc;
	// STALE: somebody else renamed it first (two inserters cannot overlap — the host serialises them,
	// embedding.c:624-629 — but an uploader can): the next attach settles it by comparing generations.
	std::lock_guard<std::mutex> lk(g_mu);
	for (Attachment &a : g_attached)
		if (a.meta == meta) a.gen = new_generation;
	return HGS_OK;
}

static int hnsw_gpu_remote_detach_impl(HnswMetadata *meta)
{
	Attachment at;
	bool found = false;
	{
		std::lock_guard<std::mutex> lk(g_mu);
		for (size_t i = 0; i < g_attached.size(); i++)
			if (g_attached[i].meta == meta)
			{
				at = g_attached[i]; found = true;
				g_attached[i] = g_attached.back();
				g_attached.pop_back();
				break;
			}
	}
	if (!found) return fail(HNSW_GPU_ERR_ARG, "meta is not attached");
	if (at.own) (void) hnsw_gpu_remote_drop(at.key);               // the private mirror of a scan that lost the upload race
	return HGS_OK;
}

// ---------------------------------------------------------------------------------------
// the drop-in symbols
// ---------------------------------------------------------------------------------------
extern "C" void hnsw_init_dist_func(void)
{
	// _PG_init (embedding.c:150) may run in the postmaster: remember the path, connect lazily in
	// the backend that searches.
	std::lock_guard<std::mutex> lk(g_mu);
	const char *env = getenv("PG_EMBEDDING_GPU_SERVER");
	if (g_path.empty() && env && *env) g_path = env;
}

static dist_t hnsw_dist_func_impl(dist_func_t dist, coord_t const *ax, coord_t const *bx, size_t dim)
{
	// One pair per call (embedding.c:1037): computed here, bit-identical to the device kernels.  The DIST
	// request stays for hosts that want the device's answer itself (the tests compare the two).
	static const bool remote = getenv("PG_EMBEDDING_GPU_REMOTE_DIST") && atoi(getenv("PG_EMBEDDING_GPU_REMOTE_DIST")) > 0;
	if (!remote) return (ax && bx) ? hostdist::dist((int) dist, ax, bx, dim) : NAN;
	hgs_hdr h, r;
	memset(&h, 0, sizeof(h));
	h.op = HGS_OP_DIST; h.aux = (uint32_t) dist; h.a0 = dim;
	if (!ax || !bx || dim == 0 || rpc(&h, ax, dim * 4, bx, dim * 4, -1, &r) != HGS_OK || t_resp.size() != sizeof(dist_t))
	{
		fprintf(stderr, "pg_embedding_amd: hnsw_dist_func failed: %s\n", t_err);
		return NAN;
	}
	dist_t out;
	memcpy(&out, t_resp.data(), sizeof(out));
	return out;
}

static bool hnsw_search_impl(HnswMetadata *meta, const coord_t *point, size_t *n_results, label_t **results)
{
	if (!meta || !point || !n_results || !results) return false;
	const size_t ef = meta->efSearch;
	Attachment at;
	bool own = false;
	if (!find_attached(meta, &at))
	{
		// No identity for this index (HnswMetadata has none, embedding.c:254): mirror it under a
		// throw-away key for this one call — always correct, O(N) per call, like the in-process shim.
		at.meta = meta; at.key = ephemeral_key(); at.gen = 1;
		if (walk_and_upload(meta, at.key, at.gen) != HGS_OK)
		{
			fprintf(stderr, "pg_embedding_amd: hnsw_search: cannot mirror the index: %s\n", t_err);
			return false;
		}
		own = true;
	}
	label_t *buf = (label_t *) malloc(ef ? ef * sizeof(label_t) : 1);        // caller frees (embedding.c:327)
	bool ok = buf != nullptr;
	size_t cnt = 0;
	if (ok && ef > 0)                    // ef = 0: searchKnn trims to zero results (hnswalg.cpp:238-240)
	{
		// Generation 0 = whatever the server holds now: the generation decided at attach time whether to
		// upload; after that another backend may have inserted (and renamed the mirror), and the newest
		// state is what this backend's own pages show too.  A mirror that was dropped meanwhile (VACUUM
		// invalidates it) is uploaded again, once.
		int rc = hnsw_gpu_remote_search(at.key, 0, point, meta->dim, ef, buf, nullptr, &cnt);
		if (rc == HGS_ERR_NOKEY && !own && walk_and_upload(meta, at.key, at.gen) == HGS_OK)
			rc = hnsw_gpu_remote_search(at.key, 0, point, meta->dim, ef, buf, nullptr, &cnt);
		ok = rc == HGS_OK;
		if (!ok) fprintf(stderr, "pg_embedding_amd: hnsw_search failed: %s\n", t_err);
	}
	if (own) (void) hnsw_gpu_remote_drop(at.key);
	if (!ok) { free(buf); return false; }
	*n_results = cnt;
	*results = buf;
	return true;
}

// hnsw_bind_point (hnswalg.cpp:279-291).  The host has already stored element `idx` zero-linked
// (embedding.c:619-621,670).  The server appends it to its mirror if the mirror is one element
// behind, runs the reference's insert in serial form and returns the changed link lists, which go
// back to the host's pages through hnsw_begin_write/hnsw_end_write: the touched neighbours first,
// then the new element (hnswalg.cpp:169-222), one write pin at a time (embedding.c:780-781).
static bool hnsw_bind_point_impl(HnswMetadata *meta, const coord_t *point, idx_t idx)
{
	if (!meta || !point) return false;
	{
		std::lock_guard<std::mutex> lk(g_mu);                      // CREATE INDEX offload: rows are only stored now,
		for (HnswMetadata *b : g_building)                         // hnsw_gpu_remote_finish_build links them all
			if (b == meta) return true;
	}
	Attachment at;
	bool own = false;
	if (!find_attached(meta, &at))
	{
		if (idx == 0) return true;                               // bindPoint links nothing, hnswalg.cpp:228
		at.meta = meta; at.key = ephemeral_key(); at.gen = 1;
		if (walk_and_upload(meta, at.key, at.gen) != HGS_OK)
		{
			fprintf(stderr, "pg_embedding_amd: hnsw_bind_point: cannot mirror the index: %s\n", t_err);
			return false;
		}
		own = true;
	}
	bool ok = false;
	do
	{
		label_t label = 0;
		if (!hnsw_begin_read(meta, idx, nullptr, nullptr, &label)) { fail(HNSW_GPU_ERR_ARG, "element %u is not stored", (unsigned) idx); break; }
		hnsw_end_read(meta);
		hgs_hdr h, r;
		memset(&h, 0, sizeof(h));
		h.op = HGS_OP_BIND; h.key = at.key; h.gen = 0; h.aux = idx; h.a0 = label;
		int rc = rpc(&h, point, meta->dim * 4, nullptr, 0, -1, &r);
		if (rc == HGS_ERR_NOKEY && !own && walk_and_upload(meta, at.key, at.gen) == HGS_OK)   // dropped meanwhile
			rc = rpc(&h, point, meta->dim * 4, nullptr, 0, -1, &r);
		if (rc != HGS_OK) break;
		const size_t rec = 1 + meta->maxM + 1;                   // [idx][count][links * maxM]
		if (t_resp.size() < 4) { fail(HGS_ERR_PROTOCOL, "bad BIND response"); break; }
		uint32_t nrec;
		memcpy(&nrec, t_resp.data(), 4);
		if (t_resp.size() != 4 + (size_t) nrec * rec * 4) { fail(HGS_ERR_PROTOCOL, "bad BIND response"); break; }
		// copy out first: a host callback may longjmp (elog(ERROR)) and t_resp is reused by the next call
		std::vector<uint32_t> recs((size_t) nrec * rec);
		if (!recs.empty()) memcpy(recs.data(), t_resp.data() + 4, recs.size() * 4);
		for (uint32_t i = 0; i < nrec; i++)
		{
			const uint32_t *p = recs.data() + (size_t) i * rec;
			idx_t *dst = nullptr;
			hnsw_begin_write(meta, p[0], &dst, nullptr, nullptr);
			memcpy(dst, p + 1, (meta->maxM + 1) * sizeof(idx_t));
			hnsw_end_write(meta);
		}
		ok = true;
	} while (0);
	if (!ok) fprintf(stderr, "pg_embedding_amd: hnsw_bind_point(%u) failed: %s\n", (unsigned) idx, t_err);
	if (own) (void) hnsw_gpu_remote_drop(at.key);
	return ok;
}

// ---------------------------------------------------------------------------------------
// The C boundary.  Nothing may unwind through it (the callers are C: embedding.c): an allocation
// failure inside the library (std::bad_alloc) becomes an ordinary failure, as the reference turns
// every exception into `false` at the same place (hnswalg.cpp:258-276, 281-290).
// ---------------------------------------------------------------------------------------
extern "C" int hnsw_gpu_remote_connect(const char *socket_path) { try { return hnsw_gpu_remote_connect_impl(socket_path); } catch (...) { return fail(HNSW_GPU_ERR_NOMEM, "out of memory"); } }
extern "C" int hnsw_gpu_remote_lookup(uint64_t key, uint64_t *generation, size_t *count, int *present) { try { return hnsw_gpu_remote_lookup_impl(key, generation, count, present); } catch (...) { return fail(HNSW_GPU_ERR_NOMEM, "out of memory"); } }
extern "C" int hnsw_gpu_remote_upload(const HnswMetadata *meta, uint64_t key, uint64_t generation, const void *elements, size_t n) { try { return hnsw_gpu_remote_upload_impl(meta, key, generation, elements, n); } catch (...) { return fail(HNSW_GPU_ERR_NOMEM, "out of memory"); } }
extern "C" int hnsw_gpu_remote_update(uint64_t key, uint64_t expected_generation, uint64_t new_generation, const HnswMetadata *meta, const void *elements, size_t first, size_t count) { try { return hnsw_gpu_remote_update_impl(key, expected_generation, new_generation, meta, elements, first, count); } catch (...) { return fail(HNSW_GPU_ERR_NOMEM, "out of memory"); } }
extern "C" int hnsw_gpu_remote_search(uint64_t key, uint64_t generation, const coord_t *query, size_t dim, size_t ef, label_t *labels, dist_t *dists, size_t *count) { try { return hnsw_gpu_remote_search_impl(key, generation, query, dim, ef, labels, dists, count); } catch (...) { return fail(HNSW_GPU_ERR_NOMEM, "out of memory"); } }
extern "C" int hnsw_gpu_remote_link(uint64_t key, size_t first, size_t count, size_t max_batch) { try { return hnsw_gpu_remote_link_impl(key, first, count, max_batch); } catch (...) { return fail(HNSW_GPU_ERR_NOMEM, "out of memory"); } }
extern "C" int hnsw_gpu_remote_export(uint64_t key, void *elements, size_t bytes) { try { return hnsw_gpu_remote_export_impl(key, elements, bytes); } catch (...) { return fail(HNSW_GPU_ERR_NOMEM, "out of memory"); } }
extern "C" int hnsw_gpu_remote_set_deleted(uint64_t key, idx_t idx, int deleted) { try { return hnsw_gpu_remote_set_deleted_impl(key, idx, deleted); } catch (...) { return fail(HNSW_GPU_ERR_NOMEM, "out of memory"); } }
extern "C" int hnsw_gpu_remote_drop(uint64_t key) { try { return hnsw_gpu_remote_drop_impl(key); } catch (...) { return fail(HNSW_GPU_ERR_NOMEM, "out of memory"); } }
extern "C" int hnsw_gpu_remote_stats(hgs_stats *out) { try { return hnsw_gpu_remote_stats_impl(out); } catch (...) { return fail(HNSW_GPU_ERR_NOMEM, "out of memory"); } }
extern "C" int hnsw_gpu_remote_attach(HnswMetadata *meta, uint64_t key, uint64_t generation) { try { return hnsw_gpu_remote_attach_impl(meta, key, generation); } catch (...) { return fail(HNSW_GPU_ERR_NOMEM, "out of memory"); } }
extern "C" int hnsw_gpu_remote_begin_build(HnswMetadata *meta) { try { return hnsw_gpu_remote_begin_build_impl(meta); } catch (...) { return fail(HNSW_GPU_ERR_NOMEM, "out of memory"); } }
extern "C" int hnsw_gpu_remote_finish_build(HnswMetadata *meta, uint64_t key, uint64_t generation, size_t n_slots, size_t max_batch) { try { return hnsw_gpu_remote_finish_build_impl(meta, key, generation, n_slots, max_batch); } catch (...) { return fail(HNSW_GPU_ERR_NOMEM, "out of memory"); } }
extern "C" int hnsw_gpu_remote_advance(HnswMetadata *meta, uint64_t new_generation) { try { return hnsw_gpu_remote_advance_impl(meta, new_generation); } catch (...) { return fail(HNSW_GPU_ERR_NOMEM, "out of memory"); } }
extern "C" int hnsw_gpu_remote_detach(HnswMetadata *meta) { try { return hnsw_gpu_remote_detach_impl(meta); } catch (...) { return fail(HNSW_GPU_ERR_NOMEM, "out of memory"); } }
extern "C" dist_t hnsw_dist_func(dist_func_t dist, coord_t const *ax, coord_t const *bx, size_t dim) { try { return hnsw_dist_func_impl(dist, ax, bx, dim); } catch (...) { return NAN; } }
extern "C" bool hnsw_search(HnswMetadata *meta, const coord_t *point, size_t *n_results, label_t **results) { try { return hnsw_search_impl(meta, point, n_results, results); } catch (...) { return false; } }
extern "C" bool hnsw_bind_point(HnswMetadata *meta, const coord_t *point, idx_t idx) { try { return hnsw_bind_point_impl(meta, point, idx); } catch (...) { return false; } }
