/*
 * oracle/pgmock/regress_mini.c — TEST INFRASTRUCTURE ONLY.
 *
 * A miniature psql session over the mini-Postgres of pgmock.h: it loads the access method exactly the
 * way the server does (_PG_init, then hnsw_handler() for the IndexAmRoutine) and turns a small command
 * language into the calls the executor would make — ambuild / aminsert / ambeginscan / amrescan /
 * amgettuple / amendscan / ambulkdelete / amvacuumcleanup / amoptions / amcostestimate — plus the SQL
 * distance functions for sequential scans.  Everything below the IndexAmRoutine is the reference's
 * unmodified embedding.c; what is linked below THAT (the four symbols of embedding.h:46-47,55-56) is
 * the thing under test.  Query results are printed the way psql prints them, so they can be compared
 * with the reference's own test/expected files.
 *
 * Commands (one per line on stdin; '#' starts a comment):
 *   seqscan on|off                          SET enable_seqscan
 *   needs_wal on|off                        RelationNeedsWAL() of the indexes created from now on.  The glue
 *                                           sets unlogged = RelationNeedsWAL() (embedding.c:241), so "off" is what
 *                                           sends its page updates through generic WAL records (a COPY of the page,
 *                                           applied on finish) — the write-back of link lists has to work on both
 *   fail_read_after N                       fault injection: the N-th page read from now raises ERROR (an I/O error
 *                                           inside a storage callback: the hot path is left by longjmp)
 *   create_table T [serial]                 CREATE TABLE T (val real[])  /  (id SERIAL PRIMARY KEY, val REAL[])
 *   insert T {a,b,c} | NULL                 one row
 *   generate T N DIM SEED                   N clustered rows of DIM values k/8 (INSERT ... SELECT from a generator)
 *   create_index T NAME l2|cos|manhattan OPTS   CREATE INDEX NAME ON T USING hnsw (val <opclass>) WITH (OPTS)
 *   select T OP {a,b,c} COLS LIMIT ; TEXT   SELECT COLS FROM T ORDER BY val OP {..} [LIMIT n]; OP is <-> <=> <~>;
 *                                           COLS is val | ctid,id | ctid ; LIMIT 0 = none; TEXT is echoed first
 *   count T ; TEXT
 *   delete_all T | delete T ROWNO           DELETE (ROWNO counts stored rows from 0)
 *   vacuum T | truncate T | drop_table T
 *   cost T NAME                             startup cost the access method reports for an ordered scan
 */
#include <math.h>
#include <time.h>

#include "pgmock.h"

extern void _PG_init(void);
extern Datum hnsw_handler(PG_FUNCTION_ARGS);
extern Datum l2_distance(PG_FUNCTION_ARGS);
extern Datum cosine_distance(PG_FUNCTION_ARGS);
extern Datum manhattan_distance(PG_FUNCTION_ARGS);
void pgmock_reset_pins(void);
extern long pgmock_fail_read_countdown;

#define ROWS_PER_HEAP_PAGE 200          /* must match table_index_build_scan in pgmock.c */
#define MAX_INDEXES 8
#define MAX_TABLES 4

typedef struct
{
	char name[32];
	bool used, serial;
	int32 next_id;
	PgmockHeap heap;
	bool *free_slot;                    /* vacuumed slots, reusable by later inserts */
	size_t nfree;                       /* how many of them there are (0 = append without looking) */
	struct { Relation rel; int op; } idx[MAX_INDEXES];
	int nidx;
} Table;

static Table g_tables[MAX_TABLES];
static IndexAmRoutine *g_am;
static bool g_seqscan = true;
static bool g_needs_wal = true;       /* RelationNeedsWAL() of the indexes created from now on */
static const PGFunction g_distfn[3] = { l2_distance, cosine_distance, manhattan_distance };
static const char *g_opname[3] = { "<->", "<=>", "<~>" };

static Table *table(const char *name)
{
	for (int i = 0; i < MAX_TABLES; i++)
		if (g_tables[i].used && strcmp(g_tables[i].name, name) == 0) return &g_tables[i];
	pgmock_error("relation \"%s\" does not exist", name);
}

static void tid_of(size_t row, ItemPointer tid)
{
	ItemPointerSet(tid, (BlockNumber) (row / ROWS_PER_HEAP_PAGE), (OffsetNumber) (row % ROWS_PER_HEAP_PAGE + 1));
}
static size_t row_of(const ItemPointerData *tid)
{
	return (size_t) ItemPointerGetBlockNumber(tid) * ROWS_PER_HEAP_PAGE + (size_t) (tid->ip_posid - 1);
}

static ArrayType *parse_array(const char *s)
{
	float4 v[4096];
	int n = 0;
	if (strcmp(s, "NULL") == 0) return NULL;
	if (*s != '{') pgmock_error("malformed array literal: \"%s\"", s);
	s++;
	while (*s && *s != '}')
	{
		char *end;
		v[n++] = strtof(s, &end);
		if (end == s || n == 4096) pgmock_error("malformed array literal");
		s = *end == ',' ? end + 1 : end;
	}
	return pgmock_make_array(v, n);
}

static void format_array(const ArrayType *a, char *out, size_t cap)
{
	size_t len = 0;
	if (!a) { out[0] = 0; return; }
	const int n = ArrayGetNItems(ARR_NDIM(a), ARR_DIMS(a));
	const float4 *v = (const float4 *) ARR_DATA_PTR(a);
	len += (size_t) snprintf(out + len, cap - len, "{");
	for (int i = 0; i < n && len < cap; i++) len += (size_t) snprintf(out + len, cap - len, i ? ",%g" : "%g", (double) v[i]);
	snprintf(out + len, cap - len, "}");
}

/* psql's aligned output for a result whose cells are already text; right[c] = right-aligned column */
static void print_result(int ncols, const char **names, const bool *right, char ***cells, size_t nrows)
{
	size_t w[4];
	for (int c = 0; c < ncols; c++)
	{
		w[c] = strlen(names[c]);
		for (size_t r = 0; r < nrows; r++)
			if (strlen(cells[r][c]) > w[c]) w[c] = strlen(cells[r][c]);
	}
	for (int c = 0; c < ncols; c++)
	{
		const size_t pad = w[c] - strlen(names[c]), left = pad / 2;
		printf("%s %*s%s%*s ", c ? "|" : "", (int) left, "", names[c], (int) (pad - left), "");
	}
	printf("\n");
	for (int c = 0; c < ncols; c++)
	{
		if (c) printf("+");
		for (size_t i = 0; i < w[c] + 2; i++) printf("-");
	}
	printf("\n");
	for (size_t r = 0; r < nrows; r++)
	{
		for (int c = 0; c < ncols; c++)
		{
			const bool last = c == ncols - 1;
			if (c) printf("|");
			if (right[c]) printf(" %*s%s", (int) w[c], cells[r][c], last ? "" : " ");
			else if (last) printf(" %s", cells[r][c]);
			else printf(" %-*s ", (int) w[c], cells[r][c]);
		}
		printf("\n");
	}
	printf(nrows == 1 ? "(1 row)\n\n" : "(%zu rows)\n\n", nrows);
}

static float4 sql_distance(int op, ArrayType *a, ArrayType *b)
{
	FunctionCallInfoBaseData fc;
	memset(&fc, 0, sizeof(fc));
	fc.args[0] = PointerGetDatum(a);
	fc.args[1] = PointerGetDatum(b);
	return DatumGetFloat4(g_distfn[op](&fc));          /* l2_distance & co: embedding.c:1040-1062 -> hnsw_dist_func */
}

static void after_am_call(const char *what)
{
	if (pgmock_pins_outstanding() != 0 || pgmock_locks_outstanding() != 0)
	{
		printf("FATAL:  %s left %d buffer pins and %d content locks behind\n", what, pgmock_pins_outstanding(),
			   pgmock_locks_outstanding());
		exit(3);
	}
}

static size_t heap_insert(Table *t, ArrayType *val)
{
	size_t slot = t->heap.n;
	for (size_t i = 0; t->nfree && i < t->heap.n; i++)
		if (t->free_slot[i]) { slot = i; t->nfree--; break; }
	if (slot == t->heap.n)
	{
		if (t->heap.n == t->heap.cap)
		{
			t->heap.cap = t->heap.cap ? 2 * t->heap.cap : 64;
			t->heap.rows = (PgmockHeapRow *) repalloc(t->heap.rows, t->heap.cap * sizeof(PgmockHeapRow));
			t->free_slot = (bool *) repalloc(t->free_slot, t->heap.cap * sizeof(bool));
		}
		t->heap.n++;
	}
	t->free_slot[slot] = false;
	t->heap.rows[slot].val = val;
	t->heap.rows[slot].dead = false;
	t->heap.rows[slot].id = t->serial ? t->next_id++ : 0;
	return slot;
}

static void cmd_insert(Table *t, const char *lit)
{
	ArrayType *val = parse_array(lit);
	const size_t row = heap_insert(t, val);
	ItemPointerData tid;
	tid_of(row, &tid);
	for (int i = 0; i < t->nidx; i++)
	{
		Datum values[1] = { PointerGetDatum(val) };
		bool isnull[1] = { val == NULL };
		IndexInfo *ii = BuildIndexInfo(t->idx[i].rel);
		g_am->aminsert(t->idx[i].rel, values, isnull, &tid, NULL, UNIQUE_CHECK_NO, false, ii);
		pfree(ii);
		after_am_call("aminsert");
	}
}

static void cmd_create_index(Table *t, const char *name, const char *opclass, const char *opts)
{
	const int op = strcmp(opclass, "l2") == 0 ? 0 : strcmp(opclass, "cos") == 0 ? 1 : strcmp(opclass, "manhattan") == 0 ? 2 : -1;
	if (op < 0 || t->nidx == MAX_INDEXES) pgmock_error("operator class \"%s\" does not exist for access method \"hnsw\"", opclass);
	bytea *parsed = g_am->amoptions(PointerGetDatum(opts), true);
	Relation rel = pgmock_create_index_relation(name, &t->heap, g_distfn[op], g_needs_wal);
	rel->rd_options = parsed;
	IndexInfo *ii = BuildIndexInfo(rel);
	IndexBuildResult *res = g_am->ambuild(NULL, rel, ii);
	pfree(ii);
	pfree(res);
	after_am_call("ambuild");
	t->idx[t->nidx].rel = rel;
	t->idx[t->nidx].op = op;
	t->nidx++;
}

/* save_index / attach_index: the page image of an index written to / read from a file, so that ONE build (e.g. the batched device
 * build through the patched glue) can be searched by several binaries (the reference's objects, the drop-in libraries) over the very
 * same pages.  The table must hold the same rows (same `generate` line) — heap TIDs are positions.  Test infrastructure only. */
static void cmd_save_index(Table *t, const char *name, const char *path)
{
	for (int i = 0; i < t->nidx; i++)
		if (strcmp(t->idx[i].rel->name, name) == 0)
		{
			Relation rel = t->idx[i].rel;
			FILE *f = fopen(path, "wb");
			if (!f) pgmock_error("could not create file \"%s\"", path);
			const uint32 hdr[2] = { 0x58444947u, rel->npages[MAIN_FORKNUM] };
			bool ok = fwrite(hdr, sizeof(hdr), 1, f) == 1;
			for (BlockNumber b = 0; ok && b < rel->npages[MAIN_FORKNUM]; b++) ok = fwrite(rel->pages[MAIN_FORKNUM][b], BLCKSZ, 1, f) == 1;
			ok = fclose(f) == 0 && ok;
			if (!ok) pgmock_error("could not write file \"%s\"", path);
			printf("SAVE %u pages\n", hdr[1]);
			return;
		}
	pgmock_error("index \"%s\" does not exist", name);
}

static void cmd_attach_index(Table *t, const char *name, const char *opclass, const char *opts, const char *path)
{
	const int op = strcmp(opclass, "l2") == 0 ? 0 : strcmp(opclass, "cos") == 0 ? 1 : strcmp(opclass, "manhattan") == 0 ? 2 : -1;
	if (op < 0 || t->nidx == MAX_INDEXES) pgmock_error("operator class \"%s\" does not exist for access method \"hnsw\"", opclass);
	FILE *f = fopen(path, "rb");
	if (!f) pgmock_error("could not open file \"%s\"", path);
	uint32 hdr[2] = { 0, 0 };
	if (fread(hdr, sizeof(hdr), 1, f) != 1 || hdr[0] != 0x58444947u) { fclose(f); pgmock_error("\"%s\" is not a saved index", path); }
	bytea *parsed = g_am->amoptions(PointerGetDatum(opts), true);
	Relation rel = pgmock_create_index_relation(name, &t->heap, g_distfn[op], g_needs_wal);
	rel->rd_options = parsed;
	rel->pages[MAIN_FORKNUM] = (char **) repalloc(rel->pages[MAIN_FORKNUM], (Size) (hdr[1] ? hdr[1] : 1) * sizeof(char *));
	rel->cappages[MAIN_FORKNUM] = hdr[1] ? hdr[1] : 1;
	for (BlockNumber b = 0; b < hdr[1]; b++)
	{
		rel->pages[MAIN_FORKNUM][b] = (char *) palloc(BLCKSZ);
		if (fread(rel->pages[MAIN_FORKNUM][b], BLCKSZ, 1, f) != 1) { fclose(f); pgmock_error("\"%s\" is truncated", path); }
		rel->npages[MAIN_FORKNUM] = b + 1;
	}
	fclose(f);
	t->idx[t->nidx].rel = rel;
	t->idx[t->nidx].op = op;
	t->nidx++;
	printf("ATTACH %u pages\n", hdr[1]);
}

typedef struct { float4 d; bool null; size_t row; } SortRow;
static int cmp_sortrow(const void *a, const void *b)
{
	const SortRow *x = (const SortRow *) a, *y = (const SortRow *) b;
	if (x->null != y->null) return x->null ? 1 : -1;                   /* NULLS LAST */
	if (!x->null && x->d != y->d) return x->d < y->d ? -1 : 1;
	return x->row < y->row ? -1 : x->row > y->row ? 1 : 0;
}

static void cmd_select(Table *t, const char *opstr, const char *lit, const char *cols, long limit)
{
	int op = -1;
	for (int i = 0; i < 3; i++) if (strcmp(opstr, g_opname[i]) == 0) op = i;
	if (op < 0) pgmock_error("operator does not exist: real[] %s real[]", opstr);
	ArrayType *q;
	if (lit[0] == '@')                     /* @ROWNO: that stored row, every component moved by 1/8 */
	{
		const size_t r = (size_t) atol(lit + 1);
		if (r >= t->heap.n || !t->heap.rows[r].val) pgmock_error("row %zu does not exist", r);
		q = DatumGetArrayTypePCopy(PointerGetDatum(t->heap.rows[r].val));
		float4 *v = (float4 *) ARR_DATA_PTR(q);
		for (int i = 0; i < ArrayGetNItems(ARR_NDIM(q), ARR_DIMS(q)); i++) v[i] += 0.125f;
	}
	else
		q = parse_array(lit);
	size_t nout = 0, cap = 64;
	size_t *rows = (size_t *) palloc(cap * sizeof(size_t));
	Relation index = NULL;
	for (int i = 0; i < t->nidx && !g_seqscan; i++)
		if (t->idx[i].op == op) index = t->idx[i].rel;                 /* the last matching index, like the newest plan choice */
	if (index)
	{
		IndexScanDesc scan = g_am->ambeginscan(index, 0, 1);
		ScanKeyData ob;
		ob.sk_flags = 0;
		ob.sk_argument = PointerGetDatum(q);
		g_am->amrescan(scan, NULL, 0, &ob, 1);
		while ((limit == 0 || (long) nout < limit) && g_am->amgettuple(scan, ForwardScanDirection))
		{
			const size_t row = row_of(&scan->xs_heaptid);
			if (row >= t->heap.n || t->heap.rows[row].dead || t->free_slot[row]) continue;     /* not visible */
			if (nout == cap) rows = (size_t *) repalloc(rows, (cap *= 2) * sizeof(size_t));
			rows[nout++] = row;
		}
		g_am->amendscan(scan);
		pfree(scan->orderByData);
		pfree(scan);
		after_am_call("index scan");
	}
	else
	{
		SortRow *s = (SortRow *) palloc((t->heap.n + 1) * sizeof(SortRow));
		size_t n = 0;
		for (size_t r = 0; r < t->heap.n; r++)
		{
			if (t->heap.rows[r].dead || t->free_slot[r]) continue;
			s[n].row = r;
			s[n].null = t->heap.rows[r].val == NULL;
			s[n].d = s[n].null ? 0 : sql_distance(op, t->heap.rows[r].val, q);
			n++;
		}
		qsort(s, n, sizeof(SortRow), cmp_sortrow);
		for (size_t i = 0; i < n && (limit == 0 || (long) i < limit); i++)
		{
			if (nout == cap) rows = (size_t *) repalloc(rows, (cap *= 2) * sizeof(size_t));
			rows[nout++] = s[i].row;
		}
		pfree(s);
	}
	/* project */
	const bool want_ctid = strstr(cols, "ctid") != NULL, want_id = strstr(cols, "id") != NULL && strcmp(cols, "ctid") != 0,
			   want_val = strstr(cols, "val") != NULL;
	const char *names[4];
	bool right[4];
	int ncols = 0;
	if (want_ctid) { names[ncols] = "ctid"; right[ncols++] = false; }
	if (want_id)   { names[ncols] = "id";   right[ncols++] = true; }
	if (want_val)  { names[ncols] = "val";  right[ncols++] = false; }
	char ***cells = (char ***) palloc((nout + 1) * sizeof(char **));
	for (size_t i = 0; i < nout; i++)
	{
		cells[i] = (char **) palloc(4 * sizeof(char *));
		int c = 0;
		if (want_ctid)
		{
			ItemPointerData tid;
			tid_of(rows[i], &tid);
			cells[i][c] = (char *) palloc(32);
			snprintf(cells[i][c++], 32, "(%u,%u)", ItemPointerGetBlockNumber(&tid), (unsigned) tid.ip_posid);
		}
		if (want_id) { cells[i][c] = (char *) palloc(16); snprintf(cells[i][c++], 16, "%d", t->heap.rows[rows[i]].id); }
		if (want_val) { cells[i][c] = (char *) palloc(65536); format_array(t->heap.rows[rows[i]].val, cells[i][c++], 65536); }
	}
	print_result(ncols, names, right, cells, nout);
	for (size_t i = 0; i < nout; i++) { for (int c = 0; c < ncols; c++) pfree(cells[i][c]); pfree(cells[i]); }
	pfree(cells);
	pfree(rows);
	pfree(q);
}

static bool vacuum_callback(ItemPointer tid, void *state)
{
	Table *t = (Table *) state;
	const size_t row = row_of(tid);
	return row >= t->heap.n || t->heap.rows[row].dead || t->free_slot[row];
}

static void cmd_vacuum(Table *t)
{
	for (int i = 0; i < t->nidx; i++)
	{
		IndexVacuumInfo info = { t->idx[i].rel };
		IndexBulkDeleteResult *st = g_am->ambulkdelete(&info, NULL, vacuum_callback, t);
		st = g_am->amvacuumcleanup(&info, st);
		if (st) pfree(st);
		after_am_call("vacuum");
	}
	for (size_t r = 0; r < t->heap.n; r++)
		if (t->heap.rows[r].dead)
		{
			if (t->heap.rows[r].val) pfree(t->heap.rows[r].val);
			t->heap.rows[r].val = NULL;
			t->heap.rows[r].dead = false;
			t->free_slot[r] = true;
			t->nfree++;
		}
	while (t->heap.n > 0 && t->free_slot[t->heap.n - 1]) { t->heap.n--; t->nfree--; }   /* truncate the empty tail */
}

static void cmd_truncate(Table *t)
{
	for (size_t r = 0; r < t->heap.n; r++)
		if (t->heap.rows[r].val) pfree(t->heap.rows[r].val);
	t->heap.n = 0;
	t->nfree = 0;
	for (int i = 0; i < t->nidx; i++)          /* new relfilenode + index_build on the empty table */
	{
		pgmock_truncate_relation(t->idx[i].rel);
		IndexInfo *ii = BuildIndexInfo(t->idx[i].rel);
		IndexBuildResult *res = g_am->ambuild(NULL, t->idx[i].rel, ii);
		pfree(ii);
		pfree(res);
		after_am_call("ambuild");
	}
}

static void cmd_cost(Table *t, const char *name)
{
	for (int i = 0; i < t->nidx; i++)
		if (strcmp(t->idx[i].rel->name, name) == 0)
		{
			RelOptInfo rel = { (double) t->heap.n };
			IndexOptInfo io = { t->idx[i].rel->rd_id, 0, &rel };
			IndexPath path = { (List *) &path, &io };
			Cost startup = 0, total = 0;
			Selectivity sel = 0;
			double corr = 0, pages = 0;
			g_am->amcostestimate(NULL, &path, 1.0, &startup, &total, &sel, &corr, &pages);
			printf("startup cost %.2f pages %.0f\n", startup, pages);
			path.indexorderbys = NULL;
			g_am->amcostestimate(NULL, &path, 1.0, &startup, &total, &sel, &corr, &pages);
			printf("without ORDER BY: %s\n", startup > 1e300 ? "never" : "allowed");
			return;
		}
	pgmock_error("index \"%s\" does not exist", name);
}

static void run(char *line)
{
	char *echo = strstr(line, " ; ");
	if (echo) { *echo = 0; echo += 3; }
	char *tok[8];
	int n = 0;
	for (char *p = strtok(line, " \t\n"); p && n < 8; p = strtok(NULL, " \t\n")) tok[n++] = p;
	if (n == 0) return;
	if (echo) printf("%s\n", echo);
	if (strcmp(tok[0], "seqscan") == 0 && n == 2) g_seqscan = strcmp(tok[1], "on") == 0;
	else if (strcmp(tok[0], "needs_wal") == 0 && n == 2) g_needs_wal = strcmp(tok[1], "on") == 0;
	else if (strcmp(tok[0], "fail_read_after") == 0 && n == 2) pgmock_fail_read_countdown = atol(tok[1]);
	else if (strcmp(tok[0], "create_table") == 0 && n >= 2)
	{
		for (int i = 0; i < MAX_TABLES; i++)
			if (!g_tables[i].used)
			{
				memset(&g_tables[i], 0, sizeof(Table));
				g_tables[i].used = true;
				g_tables[i].serial = n > 2 && strcmp(tok[2], "serial") == 0;
				g_tables[i].next_id = 1;
				snprintf(g_tables[i].name, sizeof(g_tables[i].name), "%s", tok[1]);
				return;
			}
		pgmock_error("too many tables");
	}
	else if (strcmp(tok[0], "insert") == 0 && n == 3) cmd_insert(table(tok[1]), tok[2]);
	else if (strcmp(tok[0], "generate") == 0 && n == 5)
	{
		/* N rows of DIM values k/8, k < 64, clustered: INSERT ... SELECT from a generator (rows go through
		 * the indexes like any insert) */
		Table *t = table(tok[1]);
		const long rows = atol(tok[2]), dim = atol(tok[3]);
		unsigned long long lcg = strtoull(tok[4], NULL, 0) * 2862933555777941757ull + 3037000493ull;
		if (dim < 1 || dim > 4096) pgmock_error("generate: 1..4096 dimensions");
		float4 *centre = (float4 *) palloc(64 * (Size) dim * sizeof(float4)), *v = (float4 *) palloc((Size) dim * sizeof(float4));
		for (int c = 0; c < 64; c++)
			for (long d = 0; d < dim; d++)
			{
				lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
				centre[c * dim + d] = (float4) ((lcg >> 40) % 40);
			}
		for (long r = 0; r < rows; r++)
		{
			lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
			const int c = (int) ((lcg >> 33) % 64);
			for (long d = 0; d < dim; d++)
			{
				lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
				v[d] = (centre[c * dim + d] + (float4) ((lcg >> 40) % 24)) / 8.0f;
			}
			ArrayType *val = pgmock_make_array(v, (int) dim);
			const size_t row = heap_insert(t, val);
			ItemPointerData tid;
			tid_of(row, &tid);
			for (int i = 0; i < t->nidx; i++)
			{
				Datum values[1] = { PointerGetDatum(val) };
				bool isnull[1] = { false };
				IndexInfo *ii = BuildIndexInfo(t->idx[i].rel);
				g_am->aminsert(t->idx[i].rel, values, isnull, &tid, NULL, UNIQUE_CHECK_NO, false, ii);
				pfree(ii);
			}
		}
		pfree(centre);
		pfree(v);
		after_am_call("generate");
	}
	else if (strcmp(tok[0], "create_index") == 0 && n == 5) cmd_create_index(table(tok[1]), tok[2], tok[3], tok[4]);
	else if (strcmp(tok[0], "save_index") == 0 && n == 4) cmd_save_index(table(tok[1]), tok[2], tok[3]);
	else if (strcmp(tok[0], "attach_index") == 0 && n == 6) cmd_attach_index(table(tok[1]), tok[2], tok[3], tok[4], tok[5]);
	else if (strcmp(tok[0], "select") == 0 && n == 6) cmd_select(table(tok[1]), tok[2], tok[3], tok[4], atol(tok[5]));
	else if (strcmp(tok[0], "count") == 0 && n == 2)
	{
		Table *t = table(tok[1]);
		size_t c = 0;
		for (size_t r = 0; r < t->heap.n; r++) c += (!t->heap.rows[r].dead && !t->free_slot[r]) ? 1 : 0;
		char buf[32], *cell[1] = { buf }, **rowp[1] = { cell };
		const char *names[1] = { "count" };
		const bool right[1] = { true };
		snprintf(buf, sizeof(buf), "%zu", c);
		print_result(1, names, right, rowp, 1);
	}
	else if (strcmp(tok[0], "delete_all") == 0 && n == 2)
	{
		Table *t = table(tok[1]);
		for (size_t r = 0; r < t->heap.n; r++) if (!t->free_slot[r]) t->heap.rows[r].dead = true;
	}
	else if (strcmp(tok[0], "delete") == 0 && n == 3)
	{
		Table *t = table(tok[1]);
		const size_t r = (size_t) atol(tok[2]);
		if (r < t->heap.n && !t->free_slot[r]) t->heap.rows[r].dead = true;
	}
	else if (strcmp(tok[0], "vacuum") == 0 && n == 2) cmd_vacuum(table(tok[1]));
	else if (strcmp(tok[0], "truncate") == 0 && n == 2) cmd_truncate(table(tok[1]));
	else if (strcmp(tok[0], "drop_table") == 0 && n == 2)
	{
		Table *t = table(tok[1]);
		for (int i = 0; i < t->nidx; i++) pgmock_drop_relation(t->idx[i].rel);
		for (size_t r = 0; r < t->heap.n; r++) if (t->heap.rows[r].val) pfree(t->heap.rows[r].val);
		free(t->heap.rows);
		free(t->free_slot);
		t->used = false;
	}
	else if (strcmp(tok[0], "cost") == 0 && n == 3) cmd_cost(table(tok[1]), tok[2]);
	else pgmock_error("syntax error at or near \"%s\"", tok[0]);
}

/* the in-process drop-in library's cache counters, when that is what is linked underneath (include/hnsw_gpu_shim.h) */
extern void hnsw_gpu_shim_cache_stats(uint64_t out[8]) __attribute__((weak));
extern void hnsw_gpu_shim_insert_times(uint64_t out[5]) __attribute__((weak));

int main(void)
{
	static char line[1 << 20];
	FunctionCallInfoBaseData fc;
	memset(&fc, 0, sizeof(fc));
	_PG_init();                                                     /* shared library load, embedding.c:121-151 */
	g_am = (IndexAmRoutine *) DatumGetPointer(hnsw_handler(&fc));   /* CREATE ACCESS METHOD hnsw ... HANDLER hnsw_handler */
	if (!g_am->amcanorderbyop || g_am->amgettuple == NULL) { fprintf(stderr, "unexpected access method routine\n"); return 2; }
	while (fgets(line, sizeof(line), stdin))
	{
		jmp_buf trap;
		if (line[0] == '#' || line[0] == '\n') continue;
		line[strcspn(line, "\n")] = 0;
		pgmock_error_jmp = &trap;
		if (setjmp(trap) == 0)
		{
			struct timespec t0, t1;
			const bool timed = strncmp(line, "create_index", 12) == 0 || strncmp(line, "generate", 8) == 0 ||
							   (strncmp(line, "select", 6) == 0 && getenv("PGEMB_TIME_SELECTS")) ||
							   (strncmp(line, "insert", 6) == 0 && getenv("PGEMB_TIME_INSERTS"));
			char what[64];
			snprintf(what, sizeof(what), "%.60s", line);
			clock_gettime(CLOCK_MONOTONIC, &t0);
			run(line);
			clock_gettime(CLOCK_MONOTONIC, &t1);
			if (timed)                                              /* psql's \timing, on stderr */
				fprintf(stderr, "Time: %.3f ms  %s\n", 1e3 * (double) (t1.tv_sec - t0.tv_sec) + 1e-6 * (double) (t1.tv_nsec - t0.tv_nsec), what);
		}
		else
		{
			printf("ERROR:  %s\n", pgmock_last_error);              /* transaction abort: pins and locks are dropped */
			pgmock_reset_pins();
		}
		pgmock_error_jmp = NULL;
		fflush(stdout);
	}
	if (hnsw_gpu_shim_cache_stats && getenv("PGEMB_PRINT_CACHE_STATS"))
	{
		uint64_t c[8];
		hnsw_gpu_shim_cache_stats(c);
		fprintf(stderr, "shim cache: snapshots %llu searches %llu search_rounds %llu inserts %llu insert_rounds %llu patched %llu "
				"fallbacks %llu elements_read %llu\n", (unsigned long long) c[0], (unsigned long long) c[1], (unsigned long long) c[2],
				(unsigned long long) c[3], (unsigned long long) c[4], (unsigned long long) c[5], (unsigned long long) c[6],
				(unsigned long long) c[7]);
		if (hnsw_gpu_shim_insert_times)
		{
			uint64_t t[5];
			hnsw_gpu_shim_insert_times(t);
			if (t[4])
				fprintf(stderr, "shim inserts: %llu calls, per call us: prepare (validation walk) %.1f, device insert %.1f, write-back %.1f, other %.1f\n",
						(unsigned long long) t[4], t[0] / 1e3 / t[4], t[1] / 1e3 / t[4], t[2] / 1e3 / t[4], t[3] / 1e3 / t[4]);
		}
	}
	return 0;
}
