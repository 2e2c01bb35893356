//! `src/matcher/hip.rs` of saghen/frizbee, feature `hip`: the binding a maintainer adds to use `libfrizbee_hip.so`
//! (include/frizbee_hip.h) as one more `MatcherBackend` variant.
//!
//! NOT COMPILED in this repository: the build image and the GPU boxes have no Rust toolchain (profiles/r02_box_probe.txt).
//! Everything below the `extern "C"` block is built, loaded and tested here through the same C ABI (tests/test_host_abi.py:
//! the library exports exactly the symbols the header declares; tests/test_gpu_*.py drive them on an MI355X).  The struct
//! layouts are checked against the header by tests/test_host_abi.py::test_struct_layouts_match_header on the Python side;
//! keep the two in step.
//!
//! Three ways to use the backend, cheapest integration first:
//!   1. `MatcherHip::match_list` — the `Specialized::match_list` seam (src/matcher/algo.rs:17-22).  Uploads the borrowed
//!      haystacks on EVERY call.  Measured on an MI355X box: 10 M x 32 B = 8.1 ms per call (7.7 ms of it the PCIe copy
//!      at 52 GB/s) against 3.5 ms for the crate's own 64-thread CPU path — a literal drop-in at this seam is SLOWER than
//!      the CPU crate for one query.  It exists so that the crate's tests run against the backend unchanged.
//!   2. `HipCorpus` + `MatcherHip::match_list_resident` — the list stays in HBM across queries (0.24 ms per ordered query
//!      on the same list, 0.10 ms with the result left on the device): the interactive case (`Matcher::set_pattern` on
//!      every keystroke against one file list) and the one the backend is for.
//!   3. `ShardedCorpus` + `MatcherHip::match_list_parallel_sharded` — `match_list_parallel` with one GPU per worker.
use crate::{CaseMatching, Config, Match, MatchIndices, SortStrategy, UnicodeMatching};
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
struct FzbScoring {
    match_score: u16,
    mismatch_penalty: u16,
    gap_open_penalty: u16,
    gap_extend_penalty: u16,
    prefix_bonus: u16,
    capitalization_bonus: u16,
    matching_case_bonus: u16,
    exact_match_bonus: u16,
    delimiter_bonus: u16,
}
#[repr(C)]
struct FzbConfig {
    max_typos: i32, // -1 = None
    casing: i32,
    unicode: i32,
    sort: i32,
    scoring: FzbScoring,
    pf_lanes: u16, // 0 / 0: the lane pair `Matcher::get_backend` picks on this host
    sw_lanes: u16,
    matching: i32,
}
#[repr(C)]
#[derive(Clone, Copy)]
struct FzbMatch {
    index: u32,
    score: u16,
    exact: u8,
    _pad: u8,
}
#[repr(C)]
#[derive(Clone, Copy)]
struct FzbMatchIndices {
    index: u32,
    score: u16,
    exact: u8,
    _pad: u8,
    positions_begin: u32,
    positions_len: u32,
}

const FZB_ERR_PANIC: c_int = 2;
const FZB_SHARD_BY_BYTES: c_int = 1;

#[link(name = "frizbee_hip")]
extern "C" {
    fn fzb_last_error() -> *const c_char;
    fn fzb_matcher_create(cfg: *const FzbConfig, needle: *const u8, len: usize, out: *mut *mut c_void) -> c_int;
    fn fzb_matcher_set_pattern(m: *mut c_void, needle: *const u8, len: usize) -> c_int;
    fn fzb_matcher_free(m: *mut c_void);
    fn fzb_corpus_upload(bytes: *const u8, ends: *const u64, n: usize, out: *mut *mut c_void) -> c_int;
    fn fzb_corpus_free(c: *mut c_void);
    fn fzb_match_list(m: *mut c_void, c: *const c_void, out: *mut *mut FzbMatch, out_len: *mut usize) -> c_int;
    fn fzb_match_list_into(m: *mut c_void, c: *const c_void, first: usize, count: usize, index_offset: u32, out: *mut *mut FzbMatch, out_len: *mut usize) -> c_int;
    fn fzb_matches_free(p: *mut FzbMatch);
    fn fzb_match_list_indices(m: *mut c_void, c: *const c_void, selection: *const u32, n_selection: usize, out: *mut *mut FzbMatchIndices, out_len: *mut usize,
                              out_positions: *mut *mut u32) -> c_int;
    fn fzb_match_indices_free(matches: *mut FzbMatchIndices, positions: *mut u32);
    fn fzb_corpus_upload_sharded(bytes: *const u8, ends: *const u64, n: usize, ndev: c_int, flags: c_int, out: *mut *mut c_void) -> c_int;
    fn fzb_sharded_corpus_free(sc: *mut c_void);
    fn fzb_match_list_parallel_sharded(m: *mut c_void, sc: *const c_void, out: *mut *mut FzbMatch, out_len: *mut usize) -> c_int;
    fn fzb_device_count(out: *mut c_int) -> c_int;
    fn fzb_matcher_shard_report(m: *const c_void) -> *const std::os::raw::c_char;
    // (bound by hosts that move the per-shard runs themselves / hold the list in HBM already; not used by the wrappers below)
    #[allow(dead_code)]
    fn fzb_merge_shard_runs(m: *mut c_void, dev_runs: *const *const c_void, dev_counts: *const *const u32, run_caps: *const usize, nruns: usize, stream: *mut c_void,
                            out: *mut *mut FzbMatch, out_len: *mut usize) -> c_int;
    #[allow(dead_code)]
    fn fzb_corpus_build_view(c: *mut c_void, out_built: *mut c_int) -> c_int;
    // one process per GPU: the runs travel by RCCL below the boundary (csrc/host_rccl.hip)
    fn fzb_rccl_unique_id(out_id: *mut u8) -> c_int;
    fn fzb_shard_comm_create(id: *const u8, rank: c_int, world: c_int, out: *mut *mut c_void) -> c_int;
    fn fzb_shard_comm_free(comm: *mut c_void);
    fn fzb_match_list_parallel_rccl(m: *mut c_void, shard: *const c_void, index_offset: u32, comm: *mut c_void, flags: c_int, out: *mut *mut FzbMatch,
                                    out_len: *mut usize) -> c_int;
}

/// The reference panics (`assert!`) where the ABI returns FZB_ERR_PANIC, with the same text; every other code is a backend
/// error (no device, out of memory, invalid argument) — also a panic here, because `Specialized::match_list` has no error
/// channel, but with the backend's message.  Nothing unwinds across the FFI boundary: the library returns codes.
fn check(rc: c_int) {
    if rc != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(fzb_last_error()) }.to_string_lossy().into_owned();
        if rc == FZB_ERR_PANIC {
            panic!("{msg}");
        }
        panic!("frizbee hip backend: {msg} (code {rc})");
    }
}

fn pack<H: AsRef<str>>(haystacks: &[H]) -> (Vec<u8>, Vec<u64>) {
    let mut bytes = Vec::with_capacity(haystacks.iter().map(|h| h.as_ref().len()).sum());
    let mut ends = Vec::with_capacity(haystacks.len());
    for h in haystacks {
        bytes.extend_from_slice(h.as_ref().as_bytes());
        ends.push(bytes.len() as u64);
    }
    (bytes, ends)
}

/// A haystack list resident in HBM (the `&[S]` that `match_list` borrows, uploaded once).
pub struct HipCorpus {
    handle: *mut c_void,
    len: usize,
}
impl HipCorpus {
    pub fn new<H: AsRef<str>>(haystacks: &[H]) -> Self {
        let (bytes, ends) = pack(haystacks);
        let mut handle = std::ptr::null_mut();
        check(unsafe { fzb_corpus_upload(bytes.as_ptr(), ends.as_ptr(), ends.len(), &mut handle) });
        Self { handle, len: ends.len() }
    }
    pub fn len(&self) -> usize {
        self.len
    }
}
impl Drop for HipCorpus {
    fn drop(&mut self) {
        unsafe { fzb_corpus_free(self.handle) }
    }
}

/// The list cut into contiguous shards, shard g resident on GPU g (`match_list_parallel`'s chunks, src/matcher/parallel.rs:55-63).
pub struct ShardedCorpus {
    handle: *mut c_void,
}
impl ShardedCorpus {
    /// `gpus = 0`: every visible device.  `by_bytes`: shards of equal bytes instead of equal counts (ragged lists).
    pub fn new<H: AsRef<str>>(haystacks: &[H], gpus: usize, by_bytes: bool) -> Self {
        let mut have: c_int = 0;
        check(unsafe { fzb_device_count(&mut have) });
        let ndev = if gpus == 0 { have } else { gpus as c_int };
        let (bytes, ends) = pack(haystacks);
        let mut handle = std::ptr::null_mut();
        check(unsafe { fzb_corpus_upload_sharded(bytes.as_ptr(), ends.as_ptr(), ends.len(), ndev, if by_bytes { FZB_SHARD_BY_BYTES } else { 0 }, &mut handle) });
        Self { handle }
    }
}
impl Drop for ShardedCorpus {
    fn drop(&mut self) {
        unsafe { fzb_sharded_corpus_free(self.handle) }
    }
}

pub struct MatcherHip {
    handle: *mut c_void,
}

fn c_config(config: &Config, sort: i32) -> FzbConfig {
    let s = &config.scoring;
    FzbConfig {
        max_typos: config.max_typos.map(|t| t as i32).unwrap_or(-1),
        casing: match config.casing { CaseMatching::Ignore => 0, CaseMatching::Smart => 1, CaseMatching::Respect => 2 },
        unicode: match config.unicode { UnicodeMatching::Ignore => 0, UnicodeMatching::Smart => 1, UnicodeMatching::Always => 2 },
        sort,
        scoring: FzbScoring {
            match_score: s.match_score, mismatch_penalty: s.mismatch_penalty, gap_open_penalty: s.gap_open_penalty, gap_extend_penalty: s.gap_extend_penalty,
            prefix_bonus: s.prefix_bonus, capitalization_bonus: s.capitalization_bonus, matching_case_bonus: s.matching_case_bonus,
            exact_match_bonus: s.exact_match_bonus, delimiter_bonus: s.delimiter_bonus,
        },
        pf_lanes: 0,
        sw_lanes: 0,
        matching: config.matching as i32, // Fuzzy = 0, Exact, Prefix, Suffix, Substring (declaration order, src/lib.rs:414-427)
    }
}

fn copy_out(out: *mut FzbMatch, n: usize, matches: &mut Vec<Match>) {
    // Rust's `Match` layout is unspecified (not repr(C)): copy field-wise
    matches.extend(unsafe { std::slice::from_raw_parts(out, n) }.iter().map(|m| Match { index: m.index, score: m.score, exact: m.exact != 0 }));
    unsafe { fzb_matches_free(out) };
}

impl MatcherHip {
    /// `MatcherImpl::new` (src/matcher/algo.rs:57-71).  `sort`: `Specialized::match_list` never sorts, so the backend variant is
    /// built with IndexAsc; the resident / sharded entry points below honour `config.sort` themselves.
    pub fn build(needle: &str, config: &Config) -> Self {
        let sort = match config.sort {
            SortStrategy::ScoreThenIndexAsc => 0,
            SortStrategy::ScoreThenIndexDesc => 1,
            SortStrategy::IndexAsc => 2,
            SortStrategy::IndexDesc => 3,
        };
        let cfg = c_config(config, sort);
        let mut handle = std::ptr::null_mut();
        check(unsafe { fzb_matcher_create(&cfg, needle.as_ptr(), needle.len(), &mut handle) });
        Self { handle }
    }

    /// `Matcher::set_pattern` (src/matcher/mod.rs:154-165): the device workspace is kept.
    pub fn set_pattern(&mut self, needle: &str) {
        check(unsafe { fzb_matcher_set_pattern(self.handle, needle.as_ptr(), needle.len()) });
    }

    /// `Specialized::match_list` (src/matcher/algo.rs:17-22): appends, in input order, one `Match` per prefilter-passing haystack.
    /// Uploads the list for this one call — see the module comment: correct, and slower than the CPU crate for a single query.
    pub fn match_list<H: AsRef<str>>(&mut self, haystacks: &[H], haystack_index_offset: u32, matches: &mut Vec<Match>) {
        let corpus = HipCorpus::new(haystacks);
        let (mut out, mut n) = (std::ptr::null_mut(), 0usize);
        check(unsafe { fzb_match_list_into(self.handle, corpus.handle, 0, corpus.len, haystack_index_offset, &mut out, &mut n) });
        copy_out(out, n, matches);
    }

    /// `Matcher::match_list` over a resident list: scoring, reverse / radix sort on the device, one copy of the ordered records.
    pub fn match_list_resident(&mut self, corpus: &HipCorpus) -> Vec<Match> {
        let (mut out, mut n) = (std::ptr::null_mut(), 0usize);
        check(unsafe { fzb_match_list(self.handle, corpus.handle, &mut out, &mut n) });
        let mut v = Vec::with_capacity(n);
        copy_out(out, n, &mut v);
        v
    }

    /// `Matcher::match_list_parallel(haystacks, threads)` (src/matcher/parallel.rs:18-89) with the GPUs of the node as workers:
    /// per shard the pipeline on its GPU (records in index order), the runs copied device to device into one list on the current
    /// device, ordered there once (shard order is index order, so this is `match_list`'s post-step), one copy back.  Same result as
    /// `match_list` - what the per-run sort + k-way merge of parallel.rs:66-87 returns, without a host-side merge.
    pub fn match_list_parallel_sharded(&mut self, corpus: &ShardedCorpus) -> Vec<Match> {
        let (mut out, mut n) = (std::ptr::null_mut(), 0usize);
        check(unsafe { fzb_match_list_parallel_sharded(self.handle, corpus.handle, &mut out, &mut n) });
        let mut v = Vec::with_capacity(n);
        copy_out(out, n, &mut v);
        v
    }

    /// `match_list_parallel` with ONE PROCESS PER GPU: this rank scores `shard` (its contiguous share of the list, first global index
    /// `index_offset`), the library all-gathers the run lengths and moves the runs by RCCL over xGMI to rank 0 (`to_all`: to every
    /// rank), a receiver orders the whole list once on its device.  Collective: every rank of the communicator calls it with a matcher
    /// of the same needle and config.  Returns the whole list's `match_list` result on a receiver, an empty Vec elsewhere.
    pub fn match_list_parallel_rccl(&mut self, shard: &HipCorpus, index_offset: u32, comm: &mut ShardComm, to_all: bool) -> Vec<Match> {
        let (mut out, mut n) = (std::ptr::null_mut(), 0usize);
        check(unsafe { fzb_match_list_parallel_rccl(self.handle, shard.handle, index_offset, comm.handle, to_all as c_int, &mut out, &mut n) });
        let mut v = Vec::with_capacity(n);
        copy_out(out, n, &mut v);
        v
    }

    /// How the runs of the last `match_list_parallel_sharded` reached the root device: gather form and, per shard, same device /
    /// peer access enabled (xGMI, device to device) / peer access refused (the runtime stages the copy through host memory).
    pub fn shard_report(&self) -> String {
        unsafe { std::ffi::CStr::from_ptr(fzb_matcher_shard_report(self.handle)) }.to_string_lossy().into_owned()
    }

    /// `Matcher::match_list_indices` for the listed haystacks of a resident corpus (typically the top of a `match_list` result).
    pub fn match_list_indices(&mut self, corpus: &HipCorpus, selection: &[u32]) -> Vec<MatchIndices> {
        let (mut out, mut n, mut pos) = (std::ptr::null_mut(), 0usize, std::ptr::null_mut());
        check(unsafe { fzb_match_list_indices(self.handle, corpus.handle, selection.as_ptr(), selection.len(), &mut out, &mut n, &mut pos) });
        let v = unsafe { std::slice::from_raw_parts(out, n) }
            .iter()
            .map(|m| MatchIndices {
                index: m.index,
                score: m.score,
                exact: m.exact != 0,
                indices: unsafe { std::slice::from_raw_parts(pos.add(m.positions_begin as usize), m.positions_len as usize) }.to_vec(),
            })
            .collect();
        unsafe { fzb_match_indices_free(out, pos) };
        v
    }
}
impl Drop for MatcherHip {
    fn drop(&mut self) {
        unsafe { fzb_matcher_free(self.handle) }
    }
}
// `Matcher: Send` in the reference; the handle owns device buffers and is used from one thread at a time (`&mut self`)
unsafe impl Send for MatcherHip {}


/// The communicator of the one-process-per-GPU form (`fzb_shard_comm`: an RCCL communicator, a stream and the exchange buffers on the
/// rank's current device).  `ShardComm::unique_id()` on rank 0, the 128 bytes to the other ranks by whatever started them (environment,
/// file, socket, MPI), then `ShardComm::new(&id, rank, world)` on every rank (collective).
pub struct ShardComm {
    handle: *mut c_void,
}

impl ShardComm {
    pub fn unique_id() -> [u8; 128] {
        let mut id = [0u8; 128];
        check(unsafe { fzb_rccl_unique_id(id.as_mut_ptr()) });
        id
    }
    pub fn new(id: &[u8; 128], rank: usize, world: usize) -> Self {
        let mut handle = std::ptr::null_mut();
        check(unsafe { fzb_shard_comm_create(id.as_ptr(), rank as c_int, world as c_int, &mut handle) });
        ShardComm { handle }
    }
}

impl Drop for ShardComm {
    fn drop(&mut self) {
        unsafe { fzb_shard_comm_free(self.handle) }
    }
}
