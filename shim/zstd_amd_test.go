//go:build zstd_amd && go1.21
// +build zstd_amd,go1.21

package zstd

import (
	"bytes"
	"math/rand"
	"testing"
)

// Tests of the zstd_amd hooks, in the manner of zstd_test.go / zstd_bullk_test.go: the engine's frames must be the
// vendored library's frames, byte for byte, and every frame must come back as the payload.

func amdPayloads(n int) [][]byte {
	rnd := rand.New(rand.NewSource(42))
	words := []string{"GET ", "POST ", "/api/v1/series ", "status=200 ", "status=404 ", "host=web-", "latency_ms=", "\n"}
	out := make([][]byte, n)
	for i := range out {
		var b bytes.Buffer
		for b.Len() < 1+rnd.Intn(200000) {
			b.WriteString(words[rnd.Intn(len(words))])
			if rnd.Intn(4) == 0 {
				b.WriteByte(byte('0' + rnd.Intn(10)))
			}
		}
		out[i] = b.Bytes()
	}
	out[0] = []byte{}
	return out
}

func TestAmdCompressIsVendoredCompress(t *testing.T) {
	if !amdUsable {
		t.Skip("no HIP device")
	}
	for _, level := range []int{-1, 1, 3, 5, 7} {
		for _, p := range amdPayloads(12) {
			dst := make([]byte, CompressBound(len(p)))
			got := hotCompress(dst, p, level)
			want := make([]byte, CompressBound(len(p)))
			n := vendoredCompress(want, p, level)
			if got != n || !bytes.Equal(dst[:got], want[:n]) {
				t.Fatalf("level %d, %d bytes: engine frame differs from libzstd's", level, len(p))
			}
			back := make([]byte, len(p)+1)
			if m := hotDecompress(back, dst[:got]); m != len(p) || !bytes.Equal(back[:m], p) {
				t.Fatalf("level %d, %d bytes: round trip failed (%d)", level, len(p), m)
			}
		}
	}
}

func TestAmdUnsupportedLevelFallsThrough(t *testing.T) {
	p := amdPayloads(2)[1]
	out, err := CompressLevel(nil, p, 19) // btultra2: answered by the vendored C
	if err != nil {
		t.Fatal(err)
	}
	back, err := Decompress(nil, out)
	if err != nil || !bytes.Equal(back, p) {
		t.Fatalf("round trip at level 19 failed: %v", err)
	}
}

func TestAmdBatchEqualsPerCall(t *testing.T) {
	payloads := amdPayloads(64)
	out, errs, err := CompressBatch(make([][]byte, len(payloads)), payloads, 1)
	if err != nil {
		t.Fatal(err)
	}
	for i, p := range payloads {
		want, werr := CompressLevel(nil, p, 1)
		if (errs[i] == nil) != (werr == nil) || !bytes.Equal(out[i], want) {
			t.Fatalf("payload %d: batch result differs from CompressLevel", i)
		}
	}
	back, errs, err := DecompressBatch(make([][]byte, len(out)), out)
	if err != nil {
		t.Fatal(err)
	}
	for i, p := range payloads {
		if errs[i] != nil || !bytes.Equal(back[i], p) {
			t.Fatalf("payload %d: batch round trip failed: %v", i, errs[i])
		}
	}
	_, errs, _ = DecompressBatch(make([][]byte, 1), [][]byte{{}})
	if errs[0] != ErrEmptySlice {
		t.Fatalf("empty frame: want ErrEmptySlice, got %v", errs[0])
	}
}

func TestAmdBulkBatchEqualsPerCall(t *testing.T) {
	payloads := amdPayloads(32)
	dict := bytes.Repeat([]byte("GET /api/v1/series status=200 host=web-1 latency_ms=12\n"), 40)
	p, err := NewBulkProcessor(dict, BestSpeed)
	if err != nil {
		t.Fatal(err)
	}
	out, errs, err := p.CompressBatch(make([][]byte, len(payloads)), payloads)
	if err != nil {
		t.Fatal(err)
	}
	for i, s := range payloads {
		want, werr := p.Compress(nil, s)
		if (errs[i] == nil) != (werr == nil) || !bytes.Equal(out[i], want) {
			t.Fatalf("payload %d: batch result differs from (*BulkProcessor).Compress", i)
		}
	}
	back, errs, err := p.DecompressBatch(make([][]byte, len(out)), out)
	if err != nil {
		t.Fatal(err)
	}
	for i, s := range payloads {
		if len(s) == 0 {
			continue // zstd_bulk.go:117-119: an empty destination returns early
		}
		if errs[i] != nil || !bytes.Equal(back[i], s) {
			t.Fatalf("payload %d: bulk batch round trip failed: %v", i, errs[i])
		}
	}
}
