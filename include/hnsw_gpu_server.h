/*
 * hnsw_gpu_server.h — the GPU-owning search server and its client library.
 *
 * Why it exists.  Postgres is one single-threaded process per connection
 * and every scan hands the hot path ONE query (hnsw_gettuple -> hnsw_search, embedding.c:284-343).
 * A HIP context plus a multi-GB index mirror per backend cannot work, and one query per launch is
 * latency-bound (one wavefront walking ~160 dependent hops).  So the deployment shape is:
 *
 *   hnsw_gpu_server  (one process per GPU)  owns the device, keeps the HBM mirrors keyed by a
 *                    caller-chosen (key, generation) — e.g. (relfilenode, metapage LSN) — and
 *                    coalesces the SEARCH requests of all connected backends that are waiting at
 *                    the same moment into one launch of the fused search kernel; several launches
 *                    are in flight on their own HIP streams, the kernel writes each query's result
 *                    and a completion flag straight into pinned host memory, and a backend is
 *                    answered when ITS walk has ended, not when the slowest of the batch has;
 *   libembedding_gpuc.so  exports the reference's own four symbols (embedding.h:46-47,55-56 ==
 *                    hnsw_abi.h) implemented as requests to that server.  It links no HIP and
 *                    creates no device context: a backend pays one Unix-socket round trip.
 *
 * Transport: a Unix stream socket, strictly request -> response per connection (a backend has one
 * scan in flight).  Bulk element images travel as a memfd passed with SCM_RIGHTS and are mapped by
 * the server — never copied through the socket.  All integers little-endian (same host).
 *
 * The server links libhnsw_gpu.so and nothing else computes: without a gfx950 device it refuses to
 * start (exit status 3).
 */
#ifndef PG_EMBEDDING_AMD_HNSW_GPU_SERVER_H
#define PG_EMBEDDING_AMD_HNSW_GPU_SERVER_H

/* Inside the reference's own embedding.c the types come from its embedding.h: define
 * PG_EMBEDDING_AMD_HNSW_ABI_H before including this header (integration/embedding_gpu_server.patch does). */
#include "hnsw_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ wire format */

#define HGS_MAGIC    0x31534748u          /* "HGS1" */
#define HGS_VERSION  1u
#define HGS_MAX_PAYLOAD ((uint32_t) 64 << 20)

typedef struct hgs_hdr
{
	uint32_t magic;      /* HGS_MAGIC                                             */
	uint16_t op;         /* request: HGS_OP_*; the response repeats it            */
	int16_t  status;     /* response: HGS_OK or a negative HGS_ERR_* / HNSW_GPU_ERR_* */
	uint32_t len;        /* payload bytes after this header                       */
	uint32_t aux;        /* op-specific: ef / dist_func / idx / max_batch         */
	uint64_t key;        /* which mirror                                          */
	uint64_t gen;        /* its generation (0 in a SEARCH = whatever is current)  */
	uint64_t a0, a1;     /* op-specific                                           */
} hgs_hdr;               /* 48 bytes */

enum
{
	/* op            request                                              response                         */
	HGS_OP_HELLO       = 1,  /* a0 = HGS_VERSION                               a0 = version, a1 = devices     */
	HGS_OP_LOOKUP      = 2,  /* key                                            a0 = count, a1 = 1 present/0 absent, gen = current,
	                                                                              payload = u64 content version: a server-wide
	                                                                              sequence number, new at every change; for an
	                                                                              absent key the "absent" version, new at every
	                                                                              removal of a mirror (DROP, eviction) */
	HGS_OP_UPLOAD      = 3,  /* key, gen, a0 = n, a1 = 0 or 1 + the content version of the LOOKUP this snapshot was walked
	                            after, payload = HnswMetadata, fd = memfd holding n element images (embedding.c:222-228);
	                            replaces the key's mirror — unless a1 is set and the mirror has changed since (an insert's
	                            BIND, another upload, a DROP): then HGS_ERR_STALE and nothing happens */
	HGS_OP_UPDATE      = 4,  /* key, gen = NEW generation, a0 = first, a1 = count, payload = u64 expected current
	                            generation, fd = memfd with `count` images -> hnsw_gpu_index_update_from_flat */
	HGS_OP_SEARCH      = 5,  /* key, gen (0 = any), aux = ef, a0 = 1 to get distances too, payload = dim floats
	                            -> a0 = count, payload = count labels (u64) [+ count distances (f32)]:
	                            the array hnsw_search() returns (hnswalg.cpp:256-277)                */
	HGS_OP_DROP        = 6,  /* key                                                                            */
	HGS_OP_STATS       = 7,  /*                                                payload = hgs_stats             */
	HGS_OP_DIST        = 8,  /* aux = dist_func, a0 = dim, payload = 2*dim floats  payload = one f32 (distfunc.c:171-174) */
	HGS_OP_BIND        = 9,  /* key, gen (0 = any), aux = idx, a0 = label, a1 = new generation (0 = keep),
	                            payload = dim floats: hnsw_bind_point (hnswalg.cpp:279-291) in serial mode.
	                            -> payload = u32 nrec, then nrec records [u32 idx][u32 count][u32 link*maxM]:
	                            the changed link lists, the touched neighbours first, the new element last  */
	HGS_OP_LINK        = 10, /* key, a0 = first, a1 = count, aux = max_batch (0 = default): bulk
	                            hnsw_gpu_index_link (CREATE INDEX offload)                              */
	HGS_OP_EXPORT      = 11, /* key, fd = writable memfd of count*size_data_per_element bytes -> element images */
	HGS_OP_SET_DELETED = 12, /* key, aux = idx, a0 = 0/1 (embedding.c:920-926)                                */
	HGS_OP_SETGEN      = 13, /* key, gen = NEW generation, payload = u64 expected current generation: the mirror
	                            already holds the new state (it was changed through BIND), only its name moves   */
	HGS_OP_SHM         = 14, /* fd = a memfd laid out as an hgs_shm (below): from now on this connection may POST its SEARCH
	                            requests there instead of writing them to the socket, and finds the answers there          */
	/* Row shards behind ONE front (SURVEY.md 8e mode 2 for process-per-connection hosts; round 6).  One server per GPU holds one row
	 * shard of an index under the SAME key (labels globally unique: TIDs are); the server started with --shard-peers is the FRONT that
	 * backends talk to.  Per batch of searches the front searches its own shard, asks every peer for the same queries on theirs — the
	 * peers' kernels write their (dist, label) lists straight into the front's exchange buffer, a device allocation shared through an
	 * IPC handle (hnsw_gpu_shared_alloc / _open, include/hnsw_gpu.h: peer stores over xGMI from another GPU) — and merges the lists
	 * on its device (hnsw_gpu_merge_topk_strided_dev): per-shard hnsw_search + one top-k merge, nothing staged through a host.
	 * The two operations below travel front -> peer only, on a connection of their own per dispatcher of the front. */
	HGS_OP_SHARD_ATTACH = 15, /* a0 = bytes of the exchange buffer, payload = its hnsw_gpu_ipc_handle (64 bytes): the peer maps it for
	                             this connection (a previous mapping of the connection is released)                         */
	HGS_OP_SHARD_SEARCH = 16  /* key, aux = ef, a0 = nq, a1 = byte offset of this peer's [nq][ef] labels (u64) in the attached buffer,
	                             gen = byte offset of its [nq][ef] distances (f32), payload = nq * dim floats: hnsw_search() of every
	                             query on the peer's mirror of `key`, rows padded with ~0 / +inf as hnsw_gpu_search_batch_dev pads
	                             them; answered (a0 = nq) once the peer's launch has left its device                        */
};

/* Searches through shared memory.  A backend's hnsw_search is one small request and one small answer, a thousand backends make a
 * million of each per second, and on the socket every one of them is two system calls on either side plus the wake-ups of an epoll
 * thread — measured, the CPU time of that (not the device, not the server's logic) is what a many-backend host runs out of first
 * (profiles/r4ag_stream_server_run_to_run.txt).  With HGS_OP_SHM a connection gets a mailbox of its own: the backend writes request
 * header + query into it and sets `state` = POSTED with a release store; server threads that poll the mailboxes take it (TAKEN), run
 * it exactly as a SEARCH that arrived on the socket, write response header + labels [+ distances] back and set DONE; the backend
 * spins briefly and otherwise sleeps in futex(FUTEX_WAIT) on `state` (it says so in `sleeping`, and only then does the server spend
 * a FUTEX_WAKE).  One request at a time per connection, as on the socket; everything but SEARCH stays on the socket, and so does a
 * SEARCH that does not fit the mailbox.  OPT-IN on both sides (server --shm-pollers N, backend PG_EMBEDDING_GPU_SHM=1): the pollers spin, and on
 * the CPU-capped box this was measured on the mailboxes gained nothing over the socket (profiles/r4aj_mailboxes.txt).  Layout: this header, then at offset HGS_SHM_DATA `qcap` floats of query, `rcap` labels
 * (u64), `rcap` distances (f32); the server takes the capacities from the size of the file, never from these words. */
#define HGS_SHM_MAGIC 0x4D534748u            /* "HGSM": magic of a request header that was posted through a mailbox */
#define HGS_SHM_DATA  128u
typedef struct hgs_shm
{
	uint32_t state;      /* HGS_SHM_*                                                                  */
	uint32_t sleeping;   /* backend: 1 while it sleeps in futex_wait(&state)                           */
	uint32_t qcap, rcap; /* backend's note of the capacities it laid the file out for (informational)  */
	hgs_hdr  req;        /* op = HGS_OP_SEARCH, len = dim * 4; the query follows at HGS_SHM_DATA       */
	hgs_hdr  resp;       /* as on the socket; labels / distances in their areas                        */
} hgs_shm;               /* 112 bytes */
enum { HGS_SHM_IDLE = 0, HGS_SHM_POSTED = 1, HGS_SHM_TAKEN = 2, HGS_SHM_DONE = 3 };
/* bytes of a mailbox with these capacities (qcap even) */
#define HGS_SHM_BYTES(qcap, rcap) ((size_t) HGS_SHM_DATA + (size_t) (qcap) * 4u + (size_t) (rcap) * 12u)

enum
{
	HGS_OK            =   0,
	/* -1 … -5 are the HNSW_GPU_ERR_* codes of hnsw_gpu.h, passed through */
	HGS_ERR_PROTOCOL  = -20,  /* malformed request                                 */
	HGS_ERR_NOKEY     = -21,  /* no mirror under that key                          */
	HGS_ERR_STALE     = -22,  /* the mirror's generation differs (gen = current)   */
	HGS_ERR_IO        = -23,  /* client side: cannot reach / lost the server       */
	HGS_ERR_SHUTDOWN  = -24   /* the server is stopping                            */
};

typedef struct hgs_stats
{
	uint64_t connections, connections_now;
	uint64_t searches, batches, max_batch;       /* mean batch = searches / batches */
	uint64_t search_errors;
	uint64_t uploads, upload_bytes, updates, binds, evictions;
	uint64_t mirrors, mirror_elements;
	uint64_t batch_ns;                           /* host time inside search launches, summed over dispatchers */
	uint64_t kernel_ns;                          /* device time of the search kernels (HIP events), summed     */
	uint64_t uptime_ns;
	/* where a SEARCH spends its time inside the server, summed over answered searches (streamed-completion lanes): from the
	 * request's arrival (parsed by a reader) to its launch, from the launch to the moment the dispatcher saw its completion flag
	 * (the walk + the poll), and the write of the answer; a backend's round trip minus their sum is the socket hops and its own
	 * wake-up */
	uint64_t queue_ns, walk_ns, answer_ns;
	uint64_t shm_searches;                       /* of `searches`: posted through a mailbox (HGS_OP_SHM) instead of the socket */
} hgs_stats;

/* ------------------------------------------------- client side (libembedding_gpuc.so) */

/* The library also exports hnsw_search / hnsw_bind_point / hnsw_dist_func / hnsw_init_dist_func
 * (hnsw_abi.h) and imports the host's storage callbacks like hnswalg.cpp does.  The socket path
 * comes from hnsw_gpu_remote_connect() or, lazily, from the environment variable
 * PG_EMBEDDING_GPU_SERVER.  Connections are per thread and are re-made after fork().
 * All calls return HGS_OK (0) or a negative code; hnsw_gpu_remote_last_error() has the text. */

int  hnsw_gpu_remote_connect(const char *socket_path);
void hnsw_gpu_remote_disconnect(void);
const char *hnsw_gpu_remote_last_error(void);

/* Bind `meta` (the per-scan HnswMetadata of embedding.c:254) to the server-side mirror
 * (key, generation) for the drop-in symbols.  If the server does not hold that generation the
 * host index is walked through hnsw_begin_read/hnsw_end_read (embedding.c:704-767; one pin at a
 * time, page-tail holes handled) straight into a memfd and uploaded.  Where embedding.c would
 * call it: hnsw_beginscan (embedding.c:249-262) with key = relfilenode; detach in hnsw_endscan. */
int hnsw_gpu_remote_attach(HnswMetadata *meta, uint64_t key, uint64_t generation);
int hnsw_gpu_remote_detach(HnswMetadata *meta);
/* After inserts through hnsw_bind_point() on an attached meta the server-side mirror already holds the
 * new elements; this gives that state the generation the host now computes for its index (so that the
 * next attach — in this or any other backend — finds it current instead of uploading).  Where
 * embedding.c would call it: at the end of hnsw_insert / hnsw_build (integration/embedding_gpu_server.patch). */
int hnsw_gpu_remote_advance(HnswMetadata *meta, uint64_t new_generation);

/* CREATE INDEX offload (hnsw_build, embedding.c:489-537).  After begin_build, hnsw_bind_point(meta, ...)
 * only reports success — the host has stored the row zero-linked (embedding.c:619-621,670).  finish_build
 * uploads the n_slots stored elements (numbers that do not exist, i.e. page-tail holes, become
 * vacuum-flagged placeholders), links them all on the device and writes every link list back into the
 * host's pages through hnsw_begin_write/hnsw_end_write (the caller holds the index-wide writer lock, as
 * for any hnsw_bind_point).  max_batch = 1: the reference's serial insert order, bit-identical graph;
 * 0: batched bulk build — a different, equally good graph (same degree / recall,
 * profiles/r1_build_quality.txt) in seconds instead of minutes.  The mirror stays on the server as
 * (key, generation), so the first scan attaches without an upload. */
int hnsw_gpu_remote_begin_build(HnswMetadata *meta);
int hnsw_gpu_remote_finish_build(HnswMetadata *meta, uint64_t key, uint64_t generation, size_t n_slots,
								 size_t max_batch);

/* Lower level, for hosts that manage mirrors themselves. */
int hnsw_gpu_remote_lookup(uint64_t key, uint64_t *generation, size_t *count, int *present);
int hnsw_gpu_remote_upload(const HnswMetadata *meta, uint64_t key, uint64_t generation,
						   const void *elements, size_t n);
int hnsw_gpu_remote_update(uint64_t key, uint64_t expected_generation, uint64_t new_generation,
						   const HnswMetadata *meta, const void *elements, size_t first, size_t count);
/* labels: ef values; dists: ef values or NULL */
int hnsw_gpu_remote_search(uint64_t key, uint64_t generation, const coord_t *query, size_t dim, size_t ef,
						   label_t *labels, dist_t *dists, size_t *count);
int hnsw_gpu_remote_link(uint64_t key, size_t first, size_t count, size_t max_batch);
int hnsw_gpu_remote_export(uint64_t key, void *elements, size_t bytes);
int hnsw_gpu_remote_set_deleted(uint64_t key, idx_t idx, int deleted);
int hnsw_gpu_remote_drop(uint64_t key);
int hnsw_gpu_remote_stats(hgs_stats *out);

#ifdef __cplusplus
}
#endif
#endif /* PG_EMBEDDING_AMD_HNSW_GPU_SERVER_H */
