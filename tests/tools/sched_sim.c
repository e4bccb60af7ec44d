/* TEST INFRASTRUCTURE / DIAGNOSTIC: replays the traversal event trace of tests/tools/sched_trace.py (written by the oracle) through
   the lane scheduling of the persistent traversal kernels (pathtracer_amd/csrc/mipt_persistent.h) and counts wave-level events
   under a given policy: how many 64-wide inner steps, leaf phases, refills ... a queue costs and how many lanes are active in
   each.  What a ray does is fixed (the reference's node order); the policy only decides which lane runs it when.

   gcc -O2 -o /tmp/sched_sim tests/tools/sched_sim.c && /tmp/sched_sim trace.bin [name=value ...]
     policy=0   the shipped loop: refill when >= T lanes are idle, object pass, inner phase (ends when < inner_min lanes descend
                while others wait at a leaf), leaf phase
     policy=1   ready list: a fill fetches and sets up (idle lanes + free list entries) rays at once when the list is empty and
                >= T2 lanes are idle; rays that enter the mesh go to idle lanes first, the rest to a per-wave list of CAP entries;
                a lane whose ray leaves the tree takes the next list entry at the one wave-level point behind the leaf phase
     T, inner_min, CAP, T2, waves (simulated concurrently, round-robin per outer iteration), chunk
     leaf_repeat=R  policy 1: the leaf phase repeats while >= R lanes hold a leaf (0 = never)                                  */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint32_t off, n; uint8_t rootmiss; } Ray;
typedef struct { Ray* r; size_t n, cap; } Queue;
static void q_push(Queue* q, Ray r) { if (q->n == q->cap) { q->cap = q->cap ? 2 * q->cap : 1024; q->r = realloc(q->r, q->cap * sizeof(Ray)); } q->r[q->n++] = r; }

#define MAXD 16
static Queue Q[2][MAXD];
static uint8_t* T_;

enum { IDLE, FRESH, SETTLE, ALIVE };
typedef struct { int state; Ray ray; uint32_t pos; } Lane;
typedef struct { Lane l[64]; size_t chunk_next, chunk_end; int drained, first; Ray list[256]; int rc; } Wave;

/* statistics: (events, lanes) pairs */
enum { E_INNER, E_LEAF, E_LEAFROUND, E_OBJ, E_OUTER, E_REFILL, E_TAKE, E_SETTLE, E_SPILL8, E_N };
static double ev[E_N], ln[E_N];
static const char* ename[E_N] = {"inner step", "leaf phase", "leaf round (64 packed tests)", "object pass / set-up", "outer iteration", "refill / fill", "take from list", "settle", "push at depth >= 8"};
static void count(int e, int lanes) { ev[e] += 1; ln[e] += lanes; }

static int P_policy = 0, P_T = 36, P_inner_min = 16, P_CAP = 20, P_T2 = 12, P_waves = 8, P_chunk = 512, P_leaf_repeat = 0, P_low = 0, P_leaf_max = 65, P_frac = 0, P_direct = 1;
static size_t head;

static int pull(Wave* w, int widx, size_t n) {          /* the next chunk of the queue; 0 when drained */
	size_t base;
	if (w->first) { base = (size_t)widx * P_chunk; w->first = 0; }
	else { base = head + (size_t)P_waves * P_chunk; head += P_chunk; }
	if (base >= n) { w->drained = 1; w->chunk_next = w->chunk_end = 0; return 0; }
	w->chunk_next = base; w->chunk_end = base + P_chunk < n ? base + P_chunk : n;
	return 1;
}
static inline int is_inner(const Lane* l) { return l->state == ALIVE && l->pos < l->ray.n && T_[l->ray.off + l->pos] < 0x80; }
static inline int is_leaf(const Lane* l) { return l->state == ALIVE && l->pos < l->ray.n && T_[l->ray.off + l->pos] >= 0x80; }

static void inner_phase(Wave* w) {
	for (;;) {
		int mi = 0, waiting = 0, deep = 0;
		for (int i = 0; i < 64; i++) { if (is_inner(&w->l[i])) mi++; else if (w->l[i].state == ALIVE) waiting++; }
		if (mi == 0) break;
		if (mi < P_inner_min && waiting) break;
		if (P_leaf_max < 65) { int wl = 0; for (int i = 0; i < 64; i++) if (is_leaf(&w->l[i])) wl++; if (wl >= P_leaf_max) break; }
		if (P_frac && waiting && 100 * mi < P_frac * (mi + waiting)) break;
		count(E_INNER, mi);
		for (int i = 0; i < 64; i++) if (is_inner(&w->l[i])) {
			Lane* l = &w->l[i];
			/* a push at depth >= 8: the stack entries left behind the NEXT node grow past 8 */
			if (l->pos + 1 < l->ray.n) { uint8_t nx = T_[l->ray.off + l->pos + 1], cu = T_[l->ray.off + l->pos]; if (nx < 0x80 && nx > cu && nx > 8) deep++; }
			l->pos++;
		}
		if (deep) count(E_SPILL8, deep);
	}
}
static int leaf_phase(Wave* w) {          /* returns the number of lanes that held a leaf */
	int nl = 0, tests = 0;
	for (int i = 0; i < 64; i++) if (is_leaf(&w->l[i])) { nl++; tests += T_[w->l[i].ray.off + w->l[i].pos] & 0x3f; w->l[i].pos++; }
	if (nl) { count(E_LEAF, nl); for (int b = 0; b < tests; b += 64) count(E_LEAFROUND, tests - b < 64 ? tests - b : 64); }
	return nl;
}
static void finish(Wave* w) { for (int i = 0; i < 64; i++) if (w->l[i].state == ALIVE && w->l[i].pos >= w->l[i].ray.n) w->l[i].state = SETTLE; }

/* one outer iteration of the shipped loop; returns 0 when the wave is done */
static int iter_p0(Wave* w, int widx, const Queue* q) {
	int nidle = 0, nalive = 0;
	for (int i = 0; i < 64; i++) { if (w->l[i].state == IDLE) nidle++; else if (w->l[i].state == ALIVE) nalive++; }
	count(E_OUTER, nalive);
	if (!w->drained && nidle >= P_T) {
		if (w->chunk_next >= w->chunk_end) pull(w, widx, q->n);
		if (!w->drained) {
			size_t take = w->chunk_end - w->chunk_next; if (take > (size_t)nidle) take = nidle;
			size_t k = 0;
			for (int i = 0; i < 64 && k < take; i++) if (w->l[i].state == IDLE) { w->l[i].state = FRESH; w->l[i].ray = q->r[w->chunk_next + k]; w->l[i].pos = 0; k++; }
			w->chunk_next += take;
			count(E_REFILL, (int)take);
		}
	}
	int nneed = 0, nfresh = 0, nsettle = 0;
	for (int i = 0; i < 64; i++) { if (w->l[i].state == FRESH) { nneed++; nfresh++; } else if (w->l[i].state == SETTLE) { nneed++; nsettle++; } }
	if (nneed) {
		count(E_OBJ, nneed);
		if (nsettle) count(E_SETTLE, nsettle);
		for (int i = 0; i < 64; i++) {
			Lane* l = &w->l[i];
			if (l->state == FRESH) l->state = l->ray.rootmiss ? IDLE : ALIVE;
			else if (l->state == SETTLE) l->state = IDLE;
		}
	}
	nalive = 0; for (int i = 0; i < 64; i++) if (w->l[i].state == ALIVE) nalive++;
	if (nalive == 0) return !w->drained;
	if (!w->drained && 64 - nalive >= P_T) return 1;
	inner_phase(w);
	leaf_phase(w);
	finish(w);
	return 1;
}

/* ready-list policy */
static int iter_p1(Wave* w, int widx, const Queue* q) {
	int nalive = 0;
	for (int i = 0; i < 64; i++) if (w->l[i].state == ALIVE) nalive++;
	count(E_OUTER, nalive);
	/* settle finished rays, then take from the list */
	int nsettle = 0, nidle = 0;
	for (int i = 0; i < 64; i++) if (w->l[i].state == SETTLE) { nsettle++; w->l[i].state = IDLE; }
	if (nsettle) count(E_SETTLE, nsettle);
	for (int i = 0; i < 64; i++) if (w->l[i].state == IDLE) nidle++;
	if (nidle && w->rc) {
		int take = nidle < w->rc ? nidle : w->rc, k = 0;
		for (int i = 0; i < 64 && k < take; i++) if (w->l[i].state == IDLE) { w->l[i].state = ALIVE; w->l[i].ray = w->list[k]; w->l[i].pos = 0; k++; }
		memmove(w->list, w->list + take, (w->rc - take) * sizeof(Ray)); w->rc -= take; nidle -= take;
		count(E_TAKE, take);
	}
	/* fill */
	while (!w->drained && w->rc <= P_low && nidle >= P_T2) {
		if (w->chunk_next >= w->chunk_end) { if (!pull(w, widx, q->n)) break; }
		int want = (P_direct ? nidle : 0) + (P_CAP - w->rc); if (want > 64) want = 64;
		size_t F = w->chunk_end - w->chunk_next; if (F > (size_t)want) F = want;
		count(E_REFILL, (int)F); count(E_OBJ, (int)F);
		for (size_t k = 0; k < F; k++) {
			Ray r = q->r[w->chunk_next + k];
			if (r.rootmiss) continue;                          /* settled by the lane that set it up */
			int placed = 0;
			for (int i = 0; i < 64; i++) if (w->l[i].state == IDLE) { w->l[i].state = ALIVE; w->l[i].ray = r; w->l[i].pos = 0; placed = 1; nidle--; if (!P_direct) count(E_TAKE, 0); break; }
			if (!placed) w->list[w->rc++] = r;
		}
		w->chunk_next += F;
		if (F < (size_t)want) continue;       /* chunk ended: top up from the next one */
		break;
	}
	nalive = 0; for (int i = 0; i < 64; i++) if (w->l[i].state == ALIVE) nalive++;
	if (nalive == 0) return !(w->drained && w->rc == 0);
	inner_phase(w);
	int nl = leaf_phase(w);
	while (P_leaf_repeat && nl) {
		nl = 0; for (int i = 0; i < 64; i++) if (is_leaf(&w->l[i])) nl++;
		if (nl < P_leaf_repeat) break;
		leaf_phase(w);
	}
	finish(w);
	return 1;
}

static void simulate(const Queue* q) {
	Wave* W = calloc(P_waves, sizeof(Wave));
	for (int i = 0; i < P_waves; i++) W[i].first = 1;
	head = 0;
	int live = P_waves;
	char* done = calloc(P_waves, 1);
	while (live) {
		for (int i = 0; i < P_waves; i++) if (!done[i]) {
			int go = P_policy == 0 ? iter_p0(&W[i], i, q) : iter_p1(&W[i], i, q);
			if (!go) { done[i] = 1; live--; }
		}
	}
	free(W); free(done);
}

int main(int argc, char** argv) {
	if (argc < 2) { fprintf(stderr, "usage: sched_sim trace.bin [name=value ...]\n"); return 2; }
	int kind_sel = 0, depth_sel = -1;
	for (int i = 2; i < argc; i++) {
		char* e = strchr(argv[i], '='); if (!e) continue; *e = 0; int v = atoi(e + 1);
		if (!strcmp(argv[i], "policy")) P_policy = v; else if (!strcmp(argv[i], "T")) P_T = v; else if (!strcmp(argv[i], "inner_min")) P_inner_min = v;
		else if (!strcmp(argv[i], "CAP")) P_CAP = v; else if (!strcmp(argv[i], "T2")) P_T2 = v; else if (!strcmp(argv[i], "waves")) P_waves = v;
		else if (!strcmp(argv[i], "chunk")) P_chunk = v; else if (!strcmp(argv[i], "kind")) kind_sel = v; else if (!strcmp(argv[i], "depth")) depth_sel = v;
		else if (!strcmp(argv[i], "leaf_repeat")) P_leaf_repeat = v; else if (!strcmp(argv[i], "leaf_max")) P_leaf_max = v; else if (!strcmp(argv[i], "frac")) P_frac = v; else if (!strcmp(argv[i], "direct")) P_direct = v; else if (!strcmp(argv[i], "low")) P_low = v;
		else { fprintf(stderr, "unknown parameter %s\n", argv[i]); return 2; }
	}
	FILE* f = fopen(argv[1], "rb"); if (!f) { perror(argv[1]); return 1; }
	fseek(f, 0, SEEK_END); size_t n = ftell(f); fseek(f, 0, SEEK_SET);
	T_ = malloc(n + 1); if (fread(T_, 1, n, f) != n) return 1; fclose(f);
	/* parse: 0xFD path, 0xFF kind [0xFE] events */
	int ncl = 0; size_t paths = 0;
	for (size_t p = 0; p < n;) {
		uint8_t b = T_[p];
		if (b == 0xFD) { ncl = 0; paths++; p++; continue; }
		if (b != 0xFF) { fprintf(stderr, "bad trace at %zu\n", p); return 1; }
		int kind = T_[p + 1]; p += 2;
		Ray r; r.rootmiss = 0; r.off = (uint32_t)p; r.n = 0;
		if (p < n && T_[p] == 0xFE) { r.rootmiss = 1; p++; }
		else { while (p < n && T_[p] < 0xFD) p++; r.n = (uint32_t)(p - r.off); }
		int depth = kind == 0 ? ncl++ : ncl - 1;
		if (depth < 0) depth = 0;
		if (depth >= MAXD) depth = MAXD - 1;
		if (kind == 1 && r.rootmiss) continue;      /* settled by the stage that made it */
		q_push(&Q[kind][depth], r);
	}
	double rays = 0, inner_lane = 0, leaf_lane = 0;
	for (int d = 0; d < MAXD; d++) {
		if (depth_sel >= 0 && d != depth_sel) continue;
		const Queue* q = &Q[kind_sel][d];
		if (!q->n) continue;
		rays += q->n;
		simulate(q);
	}
	(void)inner_lane; (void)leaf_lane;
	printf("trace %s: %zu paths, kind %d, %0.f rays; policy %d T %d inner_min %d CAP %d T2 %d low %d leaf_repeat %d waves %d\n", argv[1], paths, kind_sel, rays, P_policy, P_T, P_inner_min, P_CAP, P_T2, P_low, P_leaf_repeat, P_waves);
	for (int e = 0; e < E_N; e++) printf("  %-32s wave events per 1000 rays %9.2f  mean active lanes %5.1f  lane events per ray %6.2f\n", ename[e], 1000 * ev[e] / rays, ev[e] ? ln[e] / ev[e] : 0.0, ln[e] / rays);
	printf("W %.1f  (I %.1f @%.1f  L %.1f @%.1f  F %.1f @%.1f  takes %.1f)\n", 1000 * (ev[E_INNER] + 2.9 * ev[E_LEAF] + 1.9 * ev[E_REFILL] + 0.3 * ev[E_TAKE]) / rays, 1000 * ev[E_INNER] / rays, ln[E_INNER] / ev[E_INNER], 1000 * ev[E_LEAF] / rays, ln[E_LEAF] / ev[E_LEAF], 1000 * ev[E_REFILL] / rays, ln[E_REFILL] / ev[E_REFILL], 1000 * ev[E_TAKE] / rays);
	return 0;
}
