/*
 * oracle/pgmock/pgmock.c — TEST INFRASTRUCTURE ONLY.  Runtime of the mini-Postgres in pgmock.h:
 * memory, errors, float4 arrays, pages (real line-pointer arithmetic: PageAddItem fails when the
 * MAXALIGNed item plus its line pointer no longer fit, which is what produces the reference's
 * tail-of-page holes in element numbers, embedding.c:229,693), a buffer manager that counts pins and
 * refuses lock requests a real backend would deadlock on, generic WAL records that work on a copy of
 * the page and apply it on finish, reloptions, relations, index scans.
 */
#include "pgmock.h"

Oid MyDatabaseId = 5;      /* one database in the mini-Postgres */

#include <stdarg.h>

int pgmock_module_magic = 1;
jmp_buf *pgmock_error_jmp = NULL;
char pgmock_last_error[512] = "";

/* ------------------------------------------------------------------ errors / memory */
static void raise_error(void)
{
	fprintf(stderr, "ERROR:  %s\n", pgmock_last_error);
	if (pgmock_error_jmp)
		longjmp(*pgmock_error_jmp, 1);
	exit(1);
}

void pgmock_error(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(pgmock_last_error, sizeof(pgmock_last_error), fmt, ap);
	va_end(ap);
	raise_error();
	abort();
}

void pgmock_log(int level, const char *fmt, ...)
{
	va_list ap;
	(void) level;
	va_start(ap, fmt);
	vfprintf(stderr, fmt, ap);
	fputc('\n', stderr);
	va_end(ap);
}

int pgmock_errcode(int code) { (void) code; return 0; }

int pgmock_errmsg(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(pgmock_last_error, sizeof(pgmock_last_error), fmt, ap);
	va_end(ap);
	return 0;
}

void pgmock_ereport_finish(int level)
{
	if (level >= ERROR) raise_error();
	fprintf(stderr, "%s\n", pgmock_last_error);
}

void pgmock_assert_failed(const char *cond, const char *file, int line)
{
	fprintf(stderr, "TRAP: failed Assert(\"%s\"), File: \"%s\", Line: %d\n", cond, file, line);
	abort();
}

void *palloc(Size n)
{
	void *p = malloc(n ? n : 1);
	if (!p) { fprintf(stderr, "out of memory\n"); abort(); }
	return p;
}
void *palloc0(Size n) { void *p = palloc(n); memset(p, 0, n); return p; }
void *repalloc(void *p, Size n)
{
	void *q = realloc(p, n ? n : 1);
	if (!q) { fprintf(stderr, "out of memory\n"); abort(); }
	return q;
}
void pfree(void *p) { free(p); }

/* ------------------------------------------------------------------ arrays, item pointers */
int ArrayGetNItems(int ndim, const int *dims)
{
	int n = ndim > 0 ? 1 : 0;
	for (int i = 0; i < ndim; i++) n *= dims[i];
	return n;
}

ArrayType *DatumGetArrayTypePCopy(Datum d)
{
	const ArrayType *a = (const ArrayType *) DatumGetPointer(d);
	ArrayType *c = (ArrayType *) palloc((Size) a->vl_len_);
	memcpy(c, a, (Size) a->vl_len_);
	return c;
}

ArrayType *pgmock_make_array(const float4 *vals, int n)
{
	const Size hdr = MAXALIGN(sizeof(ArrayType) + 2 * sizeof(int));
	ArrayType *a = (ArrayType *) palloc0(hdr + (Size) n * sizeof(float4));
	a->vl_len_ = (int32) (hdr + (Size) n * sizeof(float4));
	a->ndim = 1;
	a->dataoffset = 0;
	a->elemtype = 700;                 /* FLOAT4OID */
	ARR_DIMS(a)[0] = n;
	ARR_DIMS(a)[1] = 1;                /* lower bound */
	memcpy(ARR_DATA_PTR(a), vals, (Size) n * sizeof(float4));
	return a;
}

int32 ItemPointerCompare(ItemPointer a, ItemPointer b)
{
	const BlockNumber ba = ItemPointerGetBlockNumber(a), bb = ItemPointerGetBlockNumber(b);
	if (ba != bb) return ba < bb ? -1 : 1;
	if (a->ip_posid != b->ip_posid) return a->ip_posid < b->ip_posid ? -1 : 1;
	return 0;
}

/* ------------------------------------------------------------------ pages */
void PageInit(Page page, Size pageSize, Size specialSize)
{
	PageHeaderData *h = (PageHeaderData *) page;
	specialSize = MAXALIGN(specialSize);
	memset(page, 0, pageSize);
	h->pd_lower = (uint16) SizeOfPageHeaderData;
	h->pd_upper = (uint16) (pageSize - specialSize);
	h->pd_special = (uint16) (pageSize - specialSize);
	h->pd_pagesize_version = (uint16) (pageSize | 4);
}

OffsetNumber PageAddItemExtended(Page page, Item item, Size size, OffsetNumber offsetNumber, int flags)
{
	PageHeaderData *h = (PageHeaderData *) page;
	(void) flags;
	if (offsetNumber != InvalidOffsetNumber)
		pgmock_error("pgmock: PageAddItem at a given offset is not supported");
	const OffsetNumber off = (OffsetNumber) (PageGetMaxOffsetNumber(page) + 1);
	const Size aligned = MAXALIGN(size);
	const int lower = h->pd_lower + (int) sizeof(ItemIdData);
	const int upper = (int) h->pd_upper - (int) aligned;
	if (lower > upper)
		return InvalidOffsetNumber;
	ItemId id = PageGetItemId(page, off);
	id->lp_off = (unsigned) upper;
	id->lp_flags = 1;                   /* LP_NORMAL */
	id->lp_len = (unsigned) size;
	memcpy(page + upper, item, size);
	h->pd_lower = (uint16) lower;
	h->pd_upper = (uint16) upper;
	return off;
}

/* ------------------------------------------------------------------ relations */
#define MAX_RELS 64
static Relation g_rels[MAX_RELS];
static Oid g_next_oid = 16384;

Relation pgmock_create_index_relation(const char *name, PgmockHeap *heap, PGFunction distproc, bool needs_wal)
{
	Relation r = (Relation) palloc0(sizeof(RelationData));
	snprintf(r->name, sizeof(r->name), "%s", name);
	r->rd_id = g_next_oid++;
	r->heap = heap;
	r->distproc.fn_addr = distproc;
	r->needs_wal = needs_wal;
	for (int i = 0; i < MAX_RELS; i++)
		if (!g_rels[i]) { g_rels[i] = r; return r; }
	pgmock_error("pgmock: too many relations");
}

void pgmock_truncate_relation(Relation rel)
{
	for (int f = 0; f < 4; f++)
	{
		for (BlockNumber b = 0; b < rel->npages[f]; b++) free(rel->pages[f][b]);
		free(rel->pages[f]);
		rel->pages[f] = NULL;
		rel->npages[f] = 0;
		rel->cappages[f] = 0;
	}
}

void pgmock_drop_relation(Relation rel)
{
	if (!rel) return;
	pgmock_truncate_relation(rel);
	for (int i = 0; i < MAX_RELS; i++)
		if (g_rels[i] == rel) g_rels[i] = NULL;
	free(rel->rd_options);
	free(rel);
}

BlockNumber RelationGetNumberOfBlocksInFork(Relation rel, ForkNumber fork) { return rel->npages[fork]; }
FmgrInfo *index_getprocinfo(Relation irel, int attnum, uint16 procnum) { (void) attnum; (void) procnum; return &irel->distproc; }

Relation index_open(Oid relationId, int lockmode)
{
	(void) lockmode;
	for (int i = 0; i < MAX_RELS; i++)
		if (g_rels[i] && g_rels[i]->rd_id == relationId) return g_rels[i];
	pgmock_error("pgmock: could not open relation with OID %u", relationId);
}
void index_close(Relation relation, int lockmode) { (void) relation; (void) lockmode; }

/* ------------------------------------------------------------------ buffer manager */
/* One backend: a "buffer" is a pin.  Locks are tracked per page so that requests a real backend
 * would block on forever (its own exclusive lock) are reported instead of silently succeeding. */
#define MAX_PINS 64
typedef struct { Relation rel; ForkNumber fork; BlockNumber blk; int lockmode; bool used; } Pin;
static Pin g_pins[MAX_PINS];

static Pin *pin_of(Buffer buf)
{
	if (buf <= 0 || buf > MAX_PINS || !g_pins[buf - 1].used)
		pgmock_error("pgmock: bad buffer %d", buf);
	return &g_pins[buf - 1];
}

int pgmock_pins_outstanding(void)
{
	int n = 0;
	for (int i = 0; i < MAX_PINS; i++) n += g_pins[i].used ? 1 : 0;
	return n;
}

void pgmock_reset_pins(void) { memset(g_pins, 0, sizeof(g_pins)); }     /* what transaction abort does */

int pgmock_locks_outstanding(void)
{
	int n = 0;
	for (int i = 0; i < MAX_PINS; i++) n += (g_pins[i].used && g_pins[i].lockmode) ? 1 : 0;
	return n;
}

long pgmock_fail_read_countdown = 0;      /* fault injection: the N-th page read from now raises ERROR */

Buffer ReadBufferExtended(Relation rel, ForkNumber fork, BlockNumber blk, ReadBufferMode mode, BufferAccessStrategy strategy)
{
	(void) mode; (void) strategy;
	if (pgmock_fail_read_countdown > 0 && --pgmock_fail_read_countdown == 0)
		pgmock_error("could not read block %u in file \"%s\": Input/output error (injected)", blk, rel->name);
	if (blk == P_NEW)
	{
		blk = rel->npages[fork];
		if (blk == rel->cappages[fork])
		{
			rel->cappages[fork] = blk ? 2 * blk : 64;
			rel->pages[fork] = (char **) repalloc(rel->pages[fork], (Size) rel->cappages[fork] * sizeof(char *));
		}
		rel->pages[fork][blk] = (char *) palloc0(BLCKSZ);
		rel->npages[fork] = blk + 1;
	}
	if (blk >= rel->npages[fork])
		pgmock_error("could not read block %u in file \"%s\": read only 0 of %d bytes", blk, rel->name, BLCKSZ);
	for (int i = 0; i < MAX_PINS; i++)
		if (!g_pins[i].used)
		{
			g_pins[i].used = true; g_pins[i].rel = rel; g_pins[i].fork = fork; g_pins[i].blk = blk; g_pins[i].lockmode = 0;
			return i + 1;
		}
	pgmock_error("pgmock: no unpinned buffers available");
}

Buffer ReadBuffer(Relation rel, BlockNumber blk) { return ReadBufferExtended(rel, MAIN_FORKNUM, blk, RBM_NORMAL, NULL); }

void LockBuffer(Buffer buf, int mode)
{
	Pin *p = pin_of(buf);
	if (mode == BUFFER_LOCK_UNLOCK) { p->lockmode = 0; return; }
	if (p->lockmode) pgmock_error("pgmock: buffer %d is already locked by this backend", buf);
	for (int i = 0; i < MAX_PINS; i++)
	{
		const Pin *o = &g_pins[i];
		if (o == p || !o->used || !o->lockmode || o->rel != p->rel || o->fork != p->fork || o->blk != p->blk) continue;
		if (mode == BUFFER_LOCK_EXCLUSIVE || o->lockmode == BUFFER_LOCK_EXCLUSIVE)
			pgmock_error("pgmock: self-deadlock: block %u of \"%s\" is already locked (mode %d) by this backend, mode %d requested",
						 p->blk, p->rel->name, o->lockmode, mode);
	}
	p->lockmode = mode;
}

void LockBufferForCleanup(Buffer buf)
{
	Pin *p = pin_of(buf);
	for (int i = 0; i < MAX_PINS; i++)
	{
		const Pin *o = &g_pins[i];
		if (o != p && o->used && o->rel == p->rel && o->fork == p->fork && o->blk == p->blk)
			pgmock_error("pgmock: cleanup lock on a page this backend has pinned twice");
	}
	LockBuffer(buf, BUFFER_LOCK_EXCLUSIVE);
}

void ReleaseBuffer(Buffer buf)
{
	Pin *p = pin_of(buf);
	if (p->lockmode) pgmock_error("pgmock: releasing buffer %d while it is locked", buf);
	p->used = false;
}

void UnlockReleaseBuffer(Buffer buf)
{
	Pin *p = pin_of(buf);
	p->lockmode = 0;
	p->used = false;
}

void MarkBufferDirty(Buffer buf)
{
	Pin *p = pin_of(buf);
	if (p->lockmode != BUFFER_LOCK_EXCLUSIVE) pgmock_error("pgmock: MarkBufferDirty without the exclusive content lock");
}

Page BufferGetPage(Buffer buf)
{
	Pin *p = pin_of(buf);
	return p->rel->pages[p->fork][p->blk];
}

BlockNumber BufferGetBlockNumber(Buffer buf) { return pin_of(buf)->blk; }

PrefetchBufferResult PrefetchBuffer(Relation rel, ForkNumber fork, BlockNumber blk)
{
	PrefetchBufferResult r = { 0 };
	(void) rel; (void) fork; (void) blk;
	return r;
}

BufferAccessStrategy GetAccessStrategy(BufferAccessStrategyType t) { (void) t; return NULL; }
void log_newpage_range(Relation rel, ForkNumber fork, BlockNumber start, BlockNumber end, bool page_std)
{ (void) rel; (void) fork; (void) start; (void) end; (void) page_std; }

/* ------------------------------------------------------------------ generic WAL: copy, then apply */
#define MAX_GENERIC_XLOG_PAGES 4
struct GenericXLogState
{
	Relation rel;
	int n;
	Buffer buf[MAX_GENERIC_XLOG_PAGES];
	char *image[MAX_GENERIC_XLOG_PAGES];
};

GenericXLogState *GenericXLogStart(Relation rel)
{
	GenericXLogState *s = (GenericXLogState *) palloc0(sizeof(*s));
	s->rel = rel;
	return s;
}

Page GenericXLogRegisterBuffer(GenericXLogState *state, Buffer buf, int flags)
{
	(void) flags;
	if (pin_of(buf)->lockmode != BUFFER_LOCK_EXCLUSIVE)
		pgmock_error("pgmock: GenericXLogRegisterBuffer without the exclusive content lock");
	for (int i = 0; i < state->n; i++)
		if (state->buf[i] == buf) return state->image[i];
	if (state->n == MAX_GENERIC_XLOG_PAGES) pgmock_error("maximum number %d of generic xlog buffers is exceeded", MAX_GENERIC_XLOG_PAGES);
	state->buf[state->n] = buf;
	state->image[state->n] = (char *) palloc(BLCKSZ);
	memcpy(state->image[state->n], BufferGetPage(buf), BLCKSZ);
	return state->image[state->n++];
}

uint64 GenericXLogFinish(GenericXLogState *state)
{
	static uint64 lsn = 1;
	for (int i = 0; i < state->n; i++)
	{
		memcpy(BufferGetPage(state->buf[i]), state->image[i], BLCKSZ);
		free(state->image[i]);
	}
	free(state);
	return lsn++;
}

void GenericXLogAbort(GenericXLogState *state)
{
	for (int i = 0; i < state->n; i++) free(state->image[i]);
	free(state);
}

/* ------------------------------------------------------------------ reloptions */
typedef struct { relopt_kind kind; char name[32]; int def, min, max; } IntOpt;
static IntOpt g_opts[64];
static int g_nopts = 0;
static relopt_kind g_next_kind = 1;

relopt_kind add_reloption_kind(void) { return g_next_kind++; }

void add_int_reloption(relopt_kind kind, const char *name, const char *desc, int default_val, int min_val, int max_val,
					   int lockmode)
{
	(void) desc; (void) lockmode;
	if (g_nopts == 64) pgmock_error("pgmock: too many reloptions");
	g_opts[g_nopts].kind = kind;
	snprintf(g_opts[g_nopts].name, sizeof(g_opts[g_nopts].name), "%s", name);
	g_opts[g_nopts].def = default_val; g_opts[g_nopts].min = min_val; g_opts[g_nopts].max = max_val;
	g_nopts++;
}

void *build_reloptions(Datum reloptions, bool validate, relopt_kind kind, Size relopt_struct_size,
					   const relopt_parse_elt *tab, int num)
{
	char *out = (char *) palloc0(relopt_struct_size);
	const char *text = (const char *) DatumGetPointer(reloptions);
	((bytea *) out)->vl_len_ = (int32) relopt_struct_size;
	for (int t = 0; t < num; t++)
		for (int i = 0; i < g_nopts; i++)
			if (g_opts[i].kind == kind && strcmp(g_opts[i].name, tab[t].optname) == 0)
				*(int *) (out + tab[t].offset) = g_opts[i].def;
	while (text && *text)
	{
		char key[32];
		long val;
		int used = 0;
		if (sscanf(text, " %31[a-z_] = %ld %n", key, &val, &used) != 2) pgmock_error("pgmock: cannot parse reloptions \"%s\"", text);
		text += used;
		if (*text == ',') text++;
		const IntOpt *o = NULL;
		for (int i = 0; i < g_nopts; i++)
			if (g_opts[i].kind == kind && strcmp(g_opts[i].name, key) == 0) o = &g_opts[i];
		if (!o) { if (validate) pgmock_error("unrecognized parameter \"%s\"", key); continue; }
		if (validate && (val < o->min || val > o->max))
			pgmock_error("value %ld out of bounds for option \"%s\"", val, key);
		for (int t = 0; t < num; t++)
			if (strcmp(tab[t].optname, key) == 0) *(int *) (out + tab[t].offset) = (int) val;
	}
	return out;
}

/* ------------------------------------------------------------------ scans, planner, build scan */
IndexScanDesc RelationGetIndexScan(Relation index, int nkeys, int norderbys)
{
	IndexScanDesc s = (IndexScanDesc) palloc0(sizeof(IndexScanDescData));
	s->indexRelation = index;
	s->numberOfKeys = nkeys;
	s->numberOfOrderBys = norderbys;
	s->keyData = nkeys > 0 ? (ScanKey) palloc0(sizeof(ScanKeyData) * (Size) nkeys) : NULL;
	s->orderByData = norderbys > 0 ? (ScanKey) palloc0(sizeof(ScanKeyData) * (Size) norderbys) : NULL;
	return s;
}

void genericcostestimate(PlannerInfo *root, IndexPath *path, double loop_count, GenericCosts *costs)
{
	(void) root; (void) path; (void) loop_count;
	costs->indexSelectivity = 1.0;
	costs->indexCorrelation = 0.0;
}

void get_tablespace_page_costs(Oid spcid, double *spc_random_page_cost, double *spc_seq_page_cost)
{
	(void) spcid;
	if (spc_random_page_cost) *spc_random_page_cost = 4.0;       /* the server's default random_page_cost */
	if (spc_seq_page_cost) *spc_seq_page_cost = 1.0;
}

IndexInfo *BuildIndexInfo(Relation index)
{
	IndexInfo *ii = (IndexInfo *) palloc0(sizeof(IndexInfo));
	(void) index;
	ii->ii_NumIndexAttrs = 1;
	return ii;
}

double table_index_build_scan(Relation table_rel, Relation index_rel, IndexInfo *index_info, bool allow_sync, bool progress,
							  IndexBuildCallback callback, void *callback_state, void *scan)
{
	PgmockHeap *heap = index_rel->heap;
	double n = 0;
	(void) table_rel; (void) index_info; (void) allow_sync; (void) progress; (void) scan;
	for (size_t i = 0; heap && i < heap->n; i++)
	{
		ItemPointerData tid;
		Datum values[1];
		bool isnull[1];
		if (heap->rows[i].dead) continue;
		ItemPointerSet(&tid, (BlockNumber) (i / 200), (OffsetNumber) (i % 200 + 1));
		values[0] = PointerGetDatum(heap->rows[i].val);
		isnull[0] = heap->rows[i].val == NULL;
		callback(index_rel, &tid, values, isnull, true, callback_state);
		n += 1;
	}
	return n;
}
