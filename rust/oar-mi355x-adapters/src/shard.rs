//! Multi-process hosts (one process per GPU, image-parallel): the pieces of `include/oar_mi355x.h`'s "multi-process hosts"
//! section a Rust launcher needs.  Pages are independent units and the crop pool is per `predict` call (ocr.rs:594-634), so a
//! rank runs the whole pipeline on its block of pages and only FINAL results travel -- there is no collective on the data path.
//! The library ships no transport: [`Mi355xOcr::predict_packed_blob`] gives a rank's results as one contiguous blob, the host moves
//! blobs with whatever it has (RCCL through its own binding, MPI, sockets), and rank 0 calls [`merge_packed`] on them in rank order.
//!
//! | what                                   | here                                  | C entry point       |
//! |----------------------------------------|---------------------------------------|---------------------|
//! | block partition of the page list       | [`shard_range`]                       | `oar_shard_range`   |
//! | a rank's results -> wire format v1     | `Mi355xOcr::predict_packed_blob`      | `oar_ocr_pack`      |
//! | blobs in rank order -> one flat result | [`merge_packed`] -> [`PackedPages`]   | `oar_packed_merge`  |

use crate::error::check;
use crate::ffi_util::slice_or_empty;
use oar_mi355x_sys as sys;
use oar_ocr_core::core::OCRError;
use std::ops::Range;

/// Rank `rank` of `world_size` owns items `[begin, end)` = `[r n / G, (r + 1) n / G)`, the remainder spread over the first ranks.
pub fn shard_range(n_items: u64, world_size: u32, rank: u32) -> Result<Range<u64>, OCRError> {
    let (mut begin, mut end) = (0u64, 0u64);
    // SAFETY: two valid out-parameters.
    let status = unsafe { sys::oar_shard_range(n_items, world_size, rank, &mut begin, &mut end) };
    check(status).map_err(|e| e.into_adapter_error("shard", format!("shard_range(n={n_items}, world={world_size}, rank={rank})")))?;
    Ok(begin..end)
}

/// The flat result of every rank's pages, in page order: page `i` owns regions `region_offsets[i]..region_offsets[i + 1]`; region `k`
/// has the quad `points[k]`, the text score `scores[k]` and the text `utf8[text_offsets[k]..text_offsets[k + 1]]`.
#[derive(Debug, Clone, Default)]
pub struct PackedPages {
    pub region_offsets: Vec<u32>,
    pub points: Vec<[[f32; 2]; 4]>,
    pub scores: Vec<f32>,
    pub text_offsets: Vec<u64>,
    pub utf8: Vec<u8>,
}

impl PackedPages {
    pub fn pages(&self) -> usize {
        self.region_offsets.len().saturating_sub(1)
    }

    pub fn text(&self, region: usize) -> std::borrow::Cow<'_, str> {
        let (a, b) = (self.text_offsets[region] as usize, self.text_offsets[region + 1] as usize);
        String::from_utf8_lossy(&self.utf8[a..b])
    }
}

struct PackedGuard(sys::oar_packed_pages);

impl Drop for PackedGuard {
    fn drop(&mut self) {
        // SAFETY: filled by oar_packed_merge (or all-zero); the free function tolerates both.
        unsafe { sys::oar_packed_pages_free(&mut self.0) }
    }
}

/// `oar_packed_merge`: the blobs of ranks `0..G` (a block partition makes concatenation in rank order = page order).  Headers and
/// offsets are validated against each blob's length; a truncated or corrupt blob is an error, not a partial result.
pub fn merge_packed(blobs: &[&[u8]]) -> Result<PackedPages, OCRError> {
    let ptrs: Vec<*const u8> = blobs.iter().map(|b| b.as_ptr()).collect();
    let lens: Vec<usize> = blobs.iter().map(|b| b.len()).collect();
    // SAFETY: an all-zero oar_packed_pages is the documented empty value.
    let mut out = PackedGuard(unsafe { std::mem::zeroed() });
    // SAFETY: two parallel arrays of blobs.len() entries, alive for the call.
    let status = unsafe { sys::oar_packed_merge(ptrs.as_ptr(), lens.as_ptr(), blobs.len() as u32, &mut out.0) };
    check(status).map_err(|e| e.into_adapter_error("shard", format!("merge of {} blob(s)", blobs.len())))?;
    let r = &out.0;
    let (n, nr) = (r.n_images as usize, r.n_regions as usize);
    // SAFETY: lengths as documented for oar_packed_pages.
    let (offsets, points, scores, text_offsets) = unsafe {
        (
            slice_or_empty(r.region_offsets, n + 1),
            slice_or_empty(r.points, nr * 8),
            slice_or_empty(r.scores, nr),
            slice_or_empty(r.text_offsets, nr + 1),
        )
    };
    let nbytes = text_offsets.last().copied().unwrap_or(0) as usize;
    // SAFETY: utf8 holds text_offsets[nr] bytes.
    let utf8 = unsafe { slice_or_empty(r.utf8 as *const u8, nbytes) };
    Ok(PackedPages {
        region_offsets: offsets.to_vec(),
        points: points.chunks_exact(8).map(|q| [[q[0], q[1]], [q[2], q[3]], [q[4], q[5]], [q[6], q[7]]]).collect(),
        scores: scores.to_vec(),
        text_offsets: text_offsets.to_vec(),
        utf8: utf8.to_vec(),
    })
}
