//go:build mi355x

package vamana

/*
#include "semadb_amd.h"
*/
import "C"

import (
	"fmt"
	"unsafe"

	"github.com/semafind/semadb/conversion"
)

// shard/vectorstore/product.go:17-18
const (
	productQuantizerCentroidDistsKey = "_productQuantizerCentroidDists"
	productQuantizerFlatCentroidsKey = "_productQuantizerFlatCentroids"
)

// loadFromBucket reads what plainPoint.ReadFrom (plain.go:125-141) and graphNode.ReadFrom (node.go:96-111)
// read lazily: 'n'+LE64(id)+'v' -> raw little-endian float32, ...'e' -> little-endian uint64 edge list; a fresh
// bucket gets its start node (setupStartNode, vamana.go:93-120).  A bucket that holds the quantizer's tables
// (product.go:80-86) also brings the store back in its fitted state with the codes kept under 'q' (:334-357).
func (v *IndexVamana) loadFromBucket() error {
	var ids, offsets, edges []uint64
	var vectors []float32
	d := int(v.parameters.VectorSize)
	offsets = append(offsets, 0)
	err := v.bucket.ForEach(func(k, val []byte) error {
		id, ok := conversion.NodeIdFromKey(k, 'v')
		if !ok {
			return nil
		}
		if len(val) != d*4 {
			return fmt.Errorf("vector of node %d has %d bytes, expected %d", id, len(val), d*4)
		}
		ids = append(ids, id)
		vectors = append(vectors, conversion.BytesToFloat32(val)...)
		if eb := v.bucket.Get(conversion.NodeKey(id, 'e')); eb != nil {
			edges = append(edges, conversion.BytesToEdgeList(eb)...)
		}
		offsets = append(offsets, uint64(len(edges)))
		return nil
	})
	if err != nil {
		return err
	}
	if len(ids) == 0 { // fresh index
		start := randomUnitVector(d)
		if rc := C.sdb_index_set_start(v.h, (*C.float)(unsafe.Pointer(&start[0])), C.SDB_MEM_HOST); rc != C.SDB_OK {
			return lastErr("could not set start point", rc)
		}
		if err := v.bucket.Put(conversion.NodeKey(STARTID, 'v'), conversion.Float32ToBytes(start)); err != nil {
			return err
		}
		return v.bucket.Put(conversion.NodeKey(STARTID, 'e'), []byte{})
	}
	if len(edges) == 0 {
		edges = append(edges, 0) // a valid pointer for cgo
	}
	rc := C.sdb_index_load(v.h, C.uint64_t(len(ids)), (*C.uint64_t)(unsafe.Pointer(&ids[0])),
		(*C.float)(unsafe.Pointer(&vectors[0])), (*C.uint64_t)(unsafe.Pointer(&offsets[0])),
		(*C.uint64_t)(unsafe.Pointer(&edges[0])), C.SDB_MEM_HOST)
	if rc != C.SDB_OK {
		return lastErr("could not load index into HBM", rc)
	}
	if v.pq == nil {
		return nil
	}
	fcb := v.bucket.Get([]byte(productQuantizerFlatCentroidsKey))
	if fcb == nil {
		return nil // not fitted yet
	}
	pp := v.parameters.Quantizer.Product
	fc := conversion.BytesToFloat32(fcb)
	if len(fc) != pp.NumSubVectors*pp.NumCentroids*(d/pp.NumSubVectors) {
		return fmt.Errorf("stored centroids have %d floats", len(fc))
	}
	if rc := C.sdb_pq_set_codebook(v.pq, (*C.float)(unsafe.Pointer(&fc[0])), C.SDB_MEM_HOST); rc != C.SDB_OK {
		return lastErr("could not load centroids", rc)
	}
	if rc := C.sdb_index_attach_pq(v.h, v.pq, nil); rc != C.SDB_OK {
		return lastErr("could not attach quantizer", rc)
	}
	var qids []uint64
	var qcodes []uint8
	for _, id := range ids {
		if qb := v.bucket.Get(conversion.NodeKey(id, 'q')); len(qb) == pp.NumSubVectors {
			qids = append(qids, id)
			qcodes = append(qcodes, qb...)
		}
	}
	if len(qids) > 0 {
		if rc := C.sdb_index_set_codes(v.h, C.uint64_t(len(qids)), (*C.uint64_t)(unsafe.Pointer(&qids[0])),
			(*C.uint8_t)(unsafe.Pointer(&qcodes[0]))); rc != C.SDB_OK {
			return lastErr("could not load centroid ids", rc)
		}
	}
	v.pqFitted = true
	return nil
}

// exportVectors copies the graph out of HBM in bucket order (sdb_index_export): ids, vectors (optional),
// CSR offsets and edges as node ids.
func (v *IndexVamana) exportVectors(withVectors bool) (ids []uint64, vecs []float32, offsets, edges []uint64, err error) {
	var nNodes, nEdges, maxId C.uint64_t
	if rc := C.sdb_index_stats(v.h, &nNodes, &nEdges, &maxId); rc != C.SDB_OK {
		return nil, nil, nil, nil, lastErr("could not read index stats", rc)
	}
	n := int(nNodes)
	ids = make([]uint64, n)
	offsets = make([]uint64, n+1)
	edges = make([]uint64, int(nEdges)+1)
	var vp *C.float
	if withVectors {
		vecs = make([]float32, n*int(v.parameters.VectorSize))
		vp = (*C.float)(unsafe.Pointer(&vecs[0]))
	}
	if rc := C.sdb_index_export(v.h, (*C.uint64_t)(unsafe.Pointer(&ids[0])), vp,
		(*C.uint64_t)(unsafe.Pointer(&offsets[0])), (*C.uint64_t)(unsafe.Pointer(&edges[0]))); rc != C.SDB_OK {
		return nil, nil, nil, nil, lastErr("could not export index", rc)
	}
	return ids, vecs, offsets, edges[:nEdges], nil
}

// flushToBucket: IndexVamana.flush (vamana.go:265-276) -- vectors to 'n<id>v', edge lists to 'n<id>e', the max
// node id; deleted nodes leave the bucket (plain.go:143-148, node.go:129-134); a fitted quantizer adds its two
// table keys (product.go:307-320) and every point's centroid ids under 'q' (:361-373).
func (v *IndexVamana) flushToBucket(deleted []uint64) error {
	for _, id := range deleted {
		for _, suffix := range []byte{'v', 'q', 'e'} {
			if err := v.bucket.Delete(conversion.NodeKey(id, suffix)); err != nil {
				return fmt.Errorf("could not delete node %d: %w", id, err)
			}
		}
	}
	ids, vecs, offsets, edges, err := v.exportVectors(true)
	if err != nil {
		return err
	}
	d := int(v.parameters.VectorSize)
	for i, id := range ids {
		if err := v.bucket.Put(conversion.NodeKey(id, 'v'), conversion.Float32ToBytes(vecs[i*d:(i+1)*d])); err != nil {
			return fmt.Errorf("could not write vector %d: %w", id, err)
		}
		if err := v.bucket.Put(conversion.NodeKey(id, 'e'), conversion.EdgeListToBytes(edges[offsets[i]:offsets[i+1]])); err != nil {
			return fmt.Errorf("could not write node %d: %w", id, err)
		}
	}
	var nNodes, nEdges, maxId C.uint64_t
	C.sdb_index_stats(v.h, &nNodes, &nEdges, &maxId)
	if err := v.bucket.Put([]byte(MAXNODEIDKEY), conversion.Uint64ToBytes(uint64(maxId))); err != nil {
		return fmt.Errorf("could not write max node id: %w", err)
	}
	if !v.pqFitted {
		return nil
	}
	pp := v.parameters.Quantizer.Product
	M, K := pp.NumSubVectors, pp.NumCentroids
	codes := make([]uint8, len(ids)*M)
	if rc := C.sdb_index_get_codes(v.h, C.uint64_t(len(ids)), (*C.uint64_t)(unsafe.Pointer(&ids[0])),
		(*C.uint8_t)(unsafe.Pointer(&codes[0]))); rc != C.SDB_OK {
		return lastErr("could not read centroid ids", rc)
	}
	for i, id := range ids {
		if err := v.bucket.Put(conversion.NodeKey(id, 'q'), codes[i*M:(i+1)*M]); err != nil {
			return fmt.Errorf("could not write centroid ids of %d: %w", id, err)
		}
	}
	fc := make([]float32, M*K*(d/M))
	cd := make([]float32, M*K*K)
	if rc := C.sdb_pq_get_codebook(v.pq, (*C.float)(unsafe.Pointer(&fc[0])), (*C.float)(unsafe.Pointer(&cd[0]))); rc != C.SDB_OK {
		return lastErr("could not read codebook", rc)
	}
	if err := v.bucket.Put([]byte(productQuantizerCentroidDistsKey), conversion.Float32ToBytes(cd)); err != nil {
		return err
	}
	return v.bucket.Put([]byte(productQuantizerFlatCentroidsKey), conversion.Float32ToBytes(fc))
}
