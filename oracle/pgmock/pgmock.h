/*
 * oracle/pgmock/pgmock.h — TEST INFRASTRUCTURE ONLY.
 *
 * A single-process stand-in for the slice of the PostgreSQL server API that the reference's glue
 * (embedding.c) is written against: pages and line pointers with the real on-page arithmetic, a
 * buffer manager that tracks pins and locks, generic WAL records with copy-then-apply semantics,
 * reloptions, the index access-method routine, index scans, arrays of float4.  It exists so that the
 * reference's embedding.c can be compiled UNMODIFIED, where it lies under /root/reference (never
 * copied), and driven through its own IndexAmRoutine by oracle/pgmock/regress_mini.c — once linked
 * with the reference's hnswalg.o + distfunc.o, once with libembedding_gpu.so, once with
 * libembedding_gpuc.so — against the reference's own pg_regress expectations (the .out files under test/expected).
 *
 * These are declarations of PostgreSQL's public extension API, restated from its documentation
 * (https://www.postgresql.org/docs/15/indexam.html, bufmgr/bufpage/generic_xlog READMEs); nothing here
 * comes from the reference repository.  Everything is deliberately minimal: one backend, no
 * transactions, no catalog.
 */
#ifndef PGMOCK_H
#define PGMOCK_H

#include <limits.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define PG_VERSION_NUM 150000
#define BLCKSZ 8192

/* ------------------------------------------------------------------ c.h / postgres.h */
typedef uintptr_t Datum;
typedef unsigned int Oid;
typedef int16_t int16;
typedef int32_t int32;
typedef int64_t int64;
typedef uint8_t uint8;
typedef uint16_t uint16;
typedef uint32_t uint32;
typedef uint64_t uint64;
typedef size_t Size;
typedef float float4;
typedef char *Pointer;
typedef double Cost;
typedef double Selectivity;
typedef struct varlena { int32 vl_len_; } bytea;

#define InvalidOid 0u
#define PGDLLEXPORT
#define PG_MODULE_MAGIC extern int pgmock_module_magic
#define lengthof(a) (sizeof(a) / sizeof((a)[0]))
#define MAXALIGN(x) (((uintptr_t) (x) + 7u) & ~(uintptr_t) 7u)
#define MemSet(p, v, n) memset((p), (v), (n))
#define Assert(c) do { if (!(c)) pgmock_assert_failed(#c, __FILE__, __LINE__); } while (0)
void pgmock_assert_failed(const char *cond, const char *file, int line) __attribute__((noreturn));

/* elog / ereport: ERROR unwinds to the driver (longjmp), like a transaction abort */
#define DEBUG1 14
#define LOG 15
#define NOTICE 18
#define WARNING 19
#define ERROR 21
#define ERRCODE_DATA_EXCEPTION 1
void pgmock_error(const char *fmt, ...) __attribute__((noreturn, format(printf, 1, 2)));
void pgmock_log(int level, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int  pgmock_errcode(int code);
int  pgmock_errmsg(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
void pgmock_ereport_finish(int level);
#define elog(level, ...) do { if ((level) >= ERROR) pgmock_error(__VA_ARGS__); else pgmock_log((level), __VA_ARGS__); } while (0)
#define errcode(c) pgmock_errcode(c)
#define errmsg(...) pgmock_errmsg(__VA_ARGS__)
#define ereport(level, rest) do { (void) (rest); pgmock_ereport_finish(level); } while (0)

void *palloc(Size n);
void *palloc0(Size n);
void *repalloc(void *p, Size n);
void  pfree(void *p);
#define pg_qsort qsort

/* ------------------------------------------------------------------ fmgr.h */
typedef struct FunctionCallInfoBaseData { Datum args[4]; bool isnull; } FunctionCallInfoBaseData;
typedef FunctionCallInfoBaseData *FunctionCallInfo;
typedef Datum (*PGFunction)(FunctionCallInfo fcinfo);
#define PG_FUNCTION_ARGS FunctionCallInfo fcinfo
#define PG_FUNCTION_INFO_V1(f) extern Datum f(PG_FUNCTION_ARGS)
typedef struct FmgrInfo { PGFunction fn_addr; } FmgrInfo;
static inline Datum Float4GetDatum(float4 x) { uint32 u; memcpy(&u, &x, 4); return (Datum) u; }
static inline float4 DatumGetFloat4(Datum d) { uint32 u = (uint32) d; float4 x; memcpy(&x, &u, 4); return x; }
#define PG_RETURN_FLOAT4(x) return Float4GetDatum(x)
#define PG_RETURN_POINTER(x) return (Datum) (uintptr_t) (x)
#define PointerGetDatum(p) ((Datum) (uintptr_t) (p))
#define DatumGetPointer(d) ((Pointer) (uintptr_t) (d))

/* ------------------------------------------------------------------ utils/array.h (float4[] only) */
typedef struct ArrayType
{
	int32 vl_len_;      /* pgmock: plain total size in bytes (no varlena header encoding) */
	int   ndim;
	int32 dataoffset;
	Oid   elemtype;
} ArrayType;
#define ARR_NDIM(a) ((a)->ndim)
#define ARR_DIMS(a) ((int *) (((char *) (a)) + sizeof(ArrayType)))
#define ARR_DATA_PTR(a) (((char *) (a)) + MAXALIGN(sizeof(ArrayType) + 2 * sizeof(int) * (a)->ndim))
int        ArrayGetNItems(int ndim, const int *dims);
ArrayType *DatumGetArrayTypePCopy(Datum d);
#define PG_GETARG_ARRAYTYPE_P(n) ((ArrayType *) DatumGetPointer(fcinfo->args[n]))
ArrayType *pgmock_make_array(const float4 *vals, int n);      /* palloc'ed one-dimensional float4[] */

/* ------------------------------------------------------------------ storage/itemptr.h, off.h, block.h */
typedef uint32 BlockNumber;
typedef uint16 OffsetNumber;
#define InvalidBlockNumber ((BlockNumber) 0xFFFFFFFF)
#define P_NEW InvalidBlockNumber
#define InvalidOffsetNumber ((OffsetNumber) 0)
#define FirstOffsetNumber ((OffsetNumber) 1)
#define OffsetNumberNext(o) ((OffsetNumber) (1 + (o)))
typedef struct BlockIdData { uint16 bi_hi, bi_lo; } BlockIdData;
typedef struct ItemPointerData { BlockIdData ip_blkid; OffsetNumber ip_posid; } ItemPointerData;   /* 6 bytes */
typedef ItemPointerData *ItemPointer;
int32 ItemPointerCompare(ItemPointer a, ItemPointer b);
static inline void ItemPointerSet(ItemPointer p, BlockNumber b, OffsetNumber o)
{ p->ip_blkid.bi_hi = (uint16) (b >> 16); p->ip_blkid.bi_lo = (uint16) b; p->ip_posid = o; }
static inline BlockNumber ItemPointerGetBlockNumber(const ItemPointerData *p)
{ return ((BlockNumber) p->ip_blkid.bi_hi << 16) | p->ip_blkid.bi_lo; }

/* ------------------------------------------------------------------ storage/bufpage.h */
typedef char *Page;
typedef Pointer Item;
typedef struct ItemIdData { unsigned lp_off : 15, lp_flags : 2, lp_len : 15; } ItemIdData;
typedef ItemIdData *ItemId;
typedef struct PageHeaderData
{
	uint64 pd_lsn;
	uint16 pd_checksum, pd_flags;
	uint16 pd_lower, pd_upper, pd_special, pd_pagesize_version;
	uint32 pd_prune_xid;
	ItemIdData pd_linp[];
} PageHeaderData;
#define SizeOfPageHeaderData (offsetof(PageHeaderData, pd_linp))      /* 24 */
void         PageInit(Page page, Size pageSize, Size specialSize);
OffsetNumber PageAddItemExtended(Page page, Item item, Size size, OffsetNumber offsetNumber, int flags);
#define PageAddItem(page, item, size, off, overwrite, is_heap) PageAddItemExtended(page, item, size, off, 0)
#define PageGetItemId(page, off) (&((PageHeaderData *) (page))->pd_linp[(off) - 1])
#define PageGetItem(page, itemId) ((Item) (((char *) (page)) + (itemId)->lp_off))
#define PageGetSpecialPointer(page) ((char *) (page) + ((PageHeaderData *) (page))->pd_special)
static inline OffsetNumber PageGetMaxOffsetNumber(Page page)
{
	const PageHeaderData *h = (const PageHeaderData *) page;
	return h->pd_lower <= SizeOfPageHeaderData ? 0 : (OffsetNumber) ((h->pd_lower - SizeOfPageHeaderData) / sizeof(ItemIdData));
}

/* ------------------------------------------------------------------ utils/rel.h, smgr.h */
typedef enum ForkNumber { MAIN_FORKNUM = 0, FSM_FORKNUM, VISIBILITYMAP_FORKNUM, INIT_FORKNUM } ForkNumber;
struct PgmockHeap;
typedef struct RelationData
{
	bytea *rd_options;          /* parsed reloptions (amoptions) */
	/* pgmock */
	Oid    rd_id;
	char   name[64];
	char **pages[4];            /* per fork */
	BlockNumber npages[4], cappages[4];
	bool   needs_wal;
	FmgrInfo distproc;          /* support procedure 1 of the operator class */
	struct PgmockHeap *heap;    /* the indexed table (index relations) */
} RelationData;
typedef RelationData *Relation;
#define RelationNeedsWAL(rel) ((rel)->needs_wal)
#define RelationGetRelid(rel) ((rel)->rd_id)
extern Oid MyDatabaseId;                 /* miscadmin.h: OID of the database this backend is connected to */
#define RelationGetSmgr(rel) (rel)
BlockNumber RelationGetNumberOfBlocksInFork(Relation rel, ForkNumber fork);
#define RelationGetNumberOfBlocks(rel) RelationGetNumberOfBlocksInFork(rel, MAIN_FORKNUM)
FmgrInfo *index_getprocinfo(Relation irel, int attnum, uint16 procnum);
Relation  index_open(Oid relationId, int lockmode);
void      index_close(Relation relation, int lockmode);
#define NoLock 0
#define AccessExclusiveLock 8

/* ------------------------------------------------------------------ storage/bufmgr.h */
typedef int Buffer;
#define InvalidBuffer 0
typedef enum { RBM_NORMAL } ReadBufferMode;
typedef enum { BAS_NORMAL, BAS_BULKREAD } BufferAccessStrategyType;
typedef void *BufferAccessStrategy;
#define BUFFER_LOCK_UNLOCK 0
#define BUFFER_LOCK_SHARE 1
#define BUFFER_LOCK_EXCLUSIVE 2
Buffer ReadBuffer(Relation rel, BlockNumber blk);
Buffer ReadBufferExtended(Relation rel, ForkNumber fork, BlockNumber blk, ReadBufferMode mode, BufferAccessStrategy strategy);
void   LockBuffer(Buffer buf, int mode);
void   LockBufferForCleanup(Buffer buf);
void   UnlockReleaseBuffer(Buffer buf);
void   ReleaseBuffer(Buffer buf);
void   MarkBufferDirty(Buffer buf);
Page   BufferGetPage(Buffer buf);
BlockNumber BufferGetBlockNumber(Buffer buf);
#define BufferGetPageSize(buf) ((Size) BLCKSZ)
typedef struct { int dummy; } PrefetchBufferResult;
PrefetchBufferResult PrefetchBuffer(Relation rel, ForkNumber fork, BlockNumber blk);
BufferAccessStrategy GetAccessStrategy(BufferAccessStrategyType t);
void log_newpage_range(Relation rel, ForkNumber fork, BlockNumber start, BlockNumber end, bool page_std);

/* ------------------------------------------------------------------ access/generic_xlog.h */
#define GENERIC_XLOG_FULL_IMAGE 1
typedef struct GenericXLogState GenericXLogState;
GenericXLogState *GenericXLogStart(Relation rel);
Page  GenericXLogRegisterBuffer(GenericXLogState *state, Buffer buf, int flags);
uint64 GenericXLogFinish(GenericXLogState *state);
void  GenericXLogAbort(GenericXLogState *state);

/* ------------------------------------------------------------------ access/reloptions.h */
typedef int relopt_kind;
typedef enum { RELOPT_TYPE_INT } relopt_type;
typedef struct { const char *optname; relopt_type opttype; int offset; } relopt_parse_elt;
relopt_kind add_reloption_kind(void);
void  add_int_reloption(relopt_kind kind, const char *name, const char *desc, int default_val, int min_val, int max_val,
						int lockmode);
void *build_reloptions(Datum reloptions, bool validate, relopt_kind kind, Size relopt_struct_size,
					   const relopt_parse_elt *tab, int num);      /* pgmock: reloptions = "k=v,k=v" C string */

/* ------------------------------------------------------------------ access/skey.h, relscan.h, sdir.h */
#define SK_ISNULL 0x0001
typedef struct ScanKeyData { int sk_flags; Datum sk_argument; } ScanKeyData;
typedef ScanKeyData *ScanKey;
typedef enum { BackwardScanDirection = -1, NoMovementScanDirection = 0, ForwardScanDirection = 1 } ScanDirection;
#define ScanDirectionIsForward(d) ((d) == ForwardScanDirection)
typedef struct IndexScanDescData
{
	Relation indexRelation;
	int      numberOfKeys, numberOfOrderBys;
	ScanKey  keyData, orderByData;
	void    *opaque;
	ItemPointerData xs_heaptid;
	bool     xs_recheckorderby;
} IndexScanDescData;
typedef IndexScanDescData *IndexScanDesc;
IndexScanDesc RelationGetIndexScan(Relation index, int nkeys, int norderbys);

/* ------------------------------------------------------------------ nodes, planner */
typedef struct List List;
typedef struct PlannerInfo PlannerInfo;
typedef struct RelOptInfo { double rows; } RelOptInfo;
typedef struct IndexOptInfo { Oid indexoid; Oid reltablespace; RelOptInfo *rel; } IndexOptInfo;
typedef struct IndexPath { List *indexorderbys; IndexOptInfo *indexinfo; } IndexPath;
typedef struct GenericCosts
{
	Cost indexStartupCost, indexTotalCost;
	Selectivity indexSelectivity;
	double indexCorrelation, numIndexPages, numIndexTuples, spc_random_page_cost, num_sa_scans;
} GenericCosts;
void genericcostestimate(PlannerInfo *root, IndexPath *path, double loop_count, GenericCosts *costs);
void get_tablespace_page_costs(Oid spcid, double *spc_random_page_cost, double *spc_seq_page_cost);
typedef struct IndexInfo { int ii_NumIndexAttrs; } IndexInfo;
IndexInfo *BuildIndexInfo(Relation index);

/* ------------------------------------------------------------------ access/tableam.h */
typedef void (*IndexBuildCallback)(Relation index, ItemPointer tid, Datum *values, bool *isnull, bool tupleIsAlive,
								   void *state);
double table_index_build_scan(Relation table_rel, Relation index_rel, IndexInfo *index_info, bool allow_sync, bool progress,
							  IndexBuildCallback callback, void *callback_state, void *scan);

/* ------------------------------------------------------------------ access/amapi.h, genam.h, commands/vacuum.h */
#define VACUUM_OPTION_PARALLEL_BULKDEL 1
typedef enum { UNIQUE_CHECK_NO } IndexUniqueCheck;
typedef struct IndexBuildResult { double heap_tuples, index_tuples; } IndexBuildResult;
typedef struct IndexVacuumInfo { Relation index; } IndexVacuumInfo;
typedef struct IndexBulkDeleteResult
{
	BlockNumber num_pages;
	bool   estimated_count;
	double num_index_tuples, tuples_removed;
	BlockNumber pages_newly_deleted, pages_deleted, pages_free;
} IndexBulkDeleteResult;
typedef bool (*IndexBulkDeleteCallback)(ItemPointer itemptr, void *state);

typedef struct IndexAmRoutine
{
	int    type;
	uint16 amstrategies, amsupport, amoptsprocnum;
	bool   amcanorder, amcanorderbyop, amcanbackward, amcanunique, amcanmulticol, amoptionalkey, amsearcharray,
		   amsearchnulls, amstorage, amclusterable, ampredlocks, amcanparallel, amcaninclude, amusemaintenanceworkmem;
	uint8  amparallelvacuumoptions;
	Oid    amkeytype;
	IndexBuildResult *(*ambuild)(Relation heap, Relation index, IndexInfo *indexInfo);
	void (*ambuildempty)(Relation index);
	bool (*aminsert)(Relation index, Datum *values, bool *isnull, ItemPointer heap_tid, Relation heap,
					 IndexUniqueCheck checkUnique, bool indexUnchanged, IndexInfo *indexInfo);
	IndexBulkDeleteResult *(*ambulkdelete)(IndexVacuumInfo *info, IndexBulkDeleteResult *stats,
										   IndexBulkDeleteCallback callback, void *callback_state);
	IndexBulkDeleteResult *(*amvacuumcleanup)(IndexVacuumInfo *info, IndexBulkDeleteResult *stats);
	void *amcanreturn;
	void (*amcostestimate)(PlannerInfo *root, IndexPath *path, double loop_count, Cost *indexStartupCost,
						   Cost *indexTotalCost, Selectivity *indexSelectivity, double *indexCorrelation, double *indexPages);
	bytea *(*amoptions)(Datum reloptions, bool validate);
	void *amproperty, *ambuildphasename;
	bool (*amvalidate)(Oid opclassoid);
	void *amadjustmembers;
	IndexScanDesc (*ambeginscan)(Relation index, int nkeys, int norderbys);
	void (*amrescan)(IndexScanDesc scan, ScanKey keys, int nkeys, ScanKey orderbys, int norderbys);
	bool (*amgettuple)(IndexScanDesc scan, ScanDirection direction);
	void *amgetbitmap;
	void (*amendscan)(IndexScanDesc scan);
	void *ammarkpos, *amrestrpos, *amestimateparallelscan, *aminitparallelscan, *amparallelrescan;
} IndexAmRoutine;
#define makeNode(T) ((T *) palloc0(sizeof(T)))

/* ------------------------------------------------------------------ the driver's side of the mock */
typedef struct PgmockHeapRow { ArrayType *val; bool dead; int32 id; } PgmockHeapRow;   /* val NULL = SQL NULL */
typedef struct PgmockHeap { PgmockHeapRow *rows; size_t n, cap; } PgmockHeap;

Relation pgmock_create_index_relation(const char *name, PgmockHeap *heap, PGFunction distproc, bool needs_wal);
void     pgmock_drop_relation(Relation rel);
void     pgmock_truncate_relation(Relation rel);
int      pgmock_pins_outstanding(void);            /* buffers pinned right now: 0 between AM calls */
int      pgmock_locks_outstanding(void);
void     pgmock_reset_pins(void);
/* error trap: returns 0 normally, 1 after an ERROR was raised inside `body` (message in pgmock_last_error) */
#include <setjmp.h>
extern jmp_buf *pgmock_error_jmp;
extern char pgmock_last_error[512];

#endif /* PGMOCK_H */
